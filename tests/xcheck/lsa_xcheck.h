/* liblsa_xcheck.so -- CROSS-CHECK LIBRARY, tests only (not part of the product library liblsa_hip.so, not part of the drop-in
 * boundary): round 1's exact block-tridiagonal LU of the RCM-ordered operator, superseded by the nested-dissection multifrontal
 * LU (lsa_ndlu_*, include/lsa_hip.h) and kept as an independent direct solver on the device that tests/test_gpu_blocklu.py
 * compares against SuperLU and the product against.  Links against liblsa_hip.so (contexts, matrices, vectors). */
#ifndef LSA_XCHECK_H
#define LSA_XCHECK_H

#include "../../include/lsa_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- exact block-tridiagonal LU: PC LU of the ST's KSP (the reference's cylinder setting, -------------------------
 * .examples/eigenvalues.py:100; Sensitivity/__init__.py:182,260) ---------------------------------------------------- */
typedef struct lsa_blu lsa_blu;
/* C must be in a banded (RCM) order.  block_size <= 0 picks max(1024, bandwidth + 1) rounded up to 256.  The Schur
 * blocks are inverted on the device (Gauss-Jordan, partial pivoting) into n * block_size resident scalars; fails with
 * LSA_ERR_OOM when they do not fit (3D meshes) and with LSA_ERR_ZERO_PIVOT on a singular block.
 * C is borrowed: its sparse off-diagonal blocks are read at every solve, so it must outlive the factorisation. */
int lsa_blu_create(lsa_ctx *ctx, const lsa_mat *C, int32_t block_size, lsa_blu **out);
void lsa_blu_destroy(lsa_blu *f);
/* x = C^-1 b (direct solve: forward + backward block sweeps replayed from a hipGraph) */
int lsa_blu_solve(lsa_ctx *ctx, lsa_blu *f, const lsa_vec *b, lsa_vec *x);
int lsa_blu_solve_time(lsa_ctx *ctx, lsa_blu *f, const lsa_vec *b, lsa_vec *x, int iters, double *avg_ms);
int lsa_blu_info(const lsa_blu *f, int32_t *block_size, int32_t *nblocks, int32_t *bandwidth, double *seconds);
/* algorithmic bytes of one lsa_blu_solve: the Schur inverses (elimination sweep: whole blocks; substitution sweep: the
 * columns that meet a non-zero), the off-block entries of C twice, the vectors -- the numerator of an achieved GB/s */
int lsa_blu_apply_bytes(const lsa_blu *f, int64_t *bytes);
/* dependent kernel launches of one lsa_blu_solve: one per pair of blocks and sweep when the couplings to the neighbouring
 * blocks are absorbed into dense operators (blocks of <= 1024 rows), two otherwise (sparse update + dense mat-vec) */
int lsa_blu_apply_launches(const lsa_blu *f, int32_t *launches);

#ifdef __cplusplus
}
#endif
#endif
