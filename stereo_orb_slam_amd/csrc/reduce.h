// reduce.h - wave-wide sums and maxima on the vector ALU (gfx950): four DPP steps inside a 16-lane row, then the two
// lane-swap instructions CDNA4 adds (v_permlane16_swap, v_permlane32_swap) across rows.  Eighteen VALU instructions per f64
// instead of six ds_bpermute round trips through the LDS crossbar (about 100 instead of 700 cycles of dependent latency -
// what the single-workgroup kernels of the solve and the epilogue of every tiled kernel are made of).
// EVERY lane returns the result; the order of the additions is fixed.
#pragma once

#include <hip/hip_runtime.h>

namespace soslam {

template <int CTRL>
__device__ __forceinline__ double mov_dpp_f64(const double x)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

constexpr int kDppXor1 = 0xB1;          // quad_perm:[1,0,3,2]
constexpr int kDppXor2 = 0x4E;          // quad_perm:[2,3,0,1]
constexpr int kDppHalfMirror = 0x141;   // lane 7 - i of each group of 8: the other quad (whose lanes all hold its sum by then)
constexpr int kDppMirror = 0x140;       // lane 15 - i of each row: the other group of 8

// v_permlane16_swap a, b exchanges the odd rows of a with the even rows of b; started from two copies of x this leaves
// a = [r0 r0 r2 r2], b = [r1 r1 r3 r3] (r = the rows of x): the pair (a, b) is (x, partner's x) or (partner's x, x) in every
// lane, and a + b adds row 2k and row 2k + 1 in the same order everywhere.  v_permlane32_swap does the same for the halves.
struct F64Pair { double a, b; };

__device__ __forceinline__ F64Pair swap_rows16(const double x)
{
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(x), __double2loint(x), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(x), __double2hiint(x), false, false);
    return F64Pair{__hiloint2double(hi[0], lo[0]), __hiloint2double(hi[1], lo[1])};
}

__device__ __forceinline__ F64Pair swap_halves32(const double x)
{
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(x), __double2loint(x), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(x), __double2hiint(x), false, false);
    return F64Pair{__hiloint2double(hi[0], lo[0]), __hiloint2double(hi[1], lo[1])};
}

// the value of lane ^ 2^K: DPP where a single control word does it (1, 2: quad permutes; 8: row rotation), the LDS crossbar otherwise
template <int K>
__device__ __forceinline__ double lane_xor_pow2(const double x)
{
    if constexpr (K == 0) return mov_dpp_f64<kDppXor1>(x);
    else if constexpr (K == 1) return mov_dpp_f64<kDppXor2>(x);
    else if constexpr (K == 3) return mov_dpp_f64<0x128>(x);   // row_ror:8
    else return __shfl_xor(x, 1 << K, 64);
}

// the sum over each aligned group of four lanes (in all four): two quad permutes
__device__ __forceinline__ double quad_sum(double x)
{
    x += mov_dpp_f64<kDppXor1>(x);
    x += mov_dpp_f64<kDppXor2>(x);
    return x;
}

// the sum over each aligned group of L = 4, 8 or 16 lanes (in all of them)
template <int L>
__device__ __forceinline__ double group_sum(double x)
{
    static_assert(L == 4 || L == 8 || L == 16, "group sizes a DPP row covers");
    x = quad_sum(x);
    if constexpr (L >= 8) x += mov_dpp_f64<kDppHalfMirror>(x);
    if constexpr (L >= 16) x += mov_dpp_f64<kDppMirror>(x);
    return x;
}

__device__ __forceinline__ double wave_sum(double x)
{
    x += mov_dpp_f64<kDppXor1>(x);
    x += mov_dpp_f64<kDppXor2>(x);
    x += mov_dpp_f64<kDppHalfMirror>(x);
    x += mov_dpp_f64<kDppMirror>(x);
    F64Pair p = swap_rows16(x);
    x = p.a + p.b;
    p = swap_halves32(x);
    return p.a + p.b;
}

__device__ __forceinline__ double wave_max(double x)
{
    x = fmax(x, mov_dpp_f64<kDppXor1>(x));
    x = fmax(x, mov_dpp_f64<kDppXor2>(x));
    x = fmax(x, mov_dpp_f64<kDppHalfMirror>(x));
    x = fmax(x, mov_dpp_f64<kDppMirror>(x));
    F64Pair p = swap_rows16(x);
    x = fmax(p.a, p.b);
    p = swap_halves32(x);
    return fmax(p.a, p.b);
}

}  // namespace soslam
