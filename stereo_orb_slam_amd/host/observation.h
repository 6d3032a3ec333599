// observation.h - stereo observation record of the host shim.
// Same fields and order as /root/reference/src/observation.h:6-17 (point id, left/right pixel coordinates,
// sigma).  sigma is carried through files and never read by the optimiser, exactly as in the reference.
#pragma once

struct Observation {
    Observation(unsigned int point_id_, float u_l_, float v_l_, float u_r_, float v_r_, float sigma_)
        : point_id(point_id_), u_l(u_l_), v_l(v_l_), u_r(u_r_), v_r(v_r_), sigma(sigma_) {}

    unsigned int point_id;
    float u_l, v_l, u_r, v_r;
    float sigma;
};
