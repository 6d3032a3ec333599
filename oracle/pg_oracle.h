#ifndef PG_ORACLE_H
#define PG_ORACLE_H
#endif
