/*
 * mik_dev.h -- DEVELOPMENT interface of libmik.so.  Not part of the drop-in boundary (include/mik.h): nothing a host of the
 * reference needs is declared here.  The knobs select kernel variants for A/B timing (scripts/) and for the tests that pin
 * every variant against the oracle (tests/test_gpu_layouts.py, test_gpu_lookahead.py); results never depend on them.
 * Every mik_ctx carries its OWN table (a copy of the defaults when it is created); launches and object creation read only the
 * table of their context.  mik_ctx_set_tuning changes one context; mik_set_tuning is the process-wide convenience of the test
 * suite: it writes the defaults and every live context, and must not race with calls on those contexts.  A host that wants an
 * operator on its plain CSR arrays uses mik_csr_set_layout (include/mik.h), not a knob.
 */
#include "mik.h"
#ifndef MIK_DEV_H
#define MIK_DEV_H
#ifdef __cplusplus
extern "C" {
#endif
/* key / value table (all default to 0; results never depend on them):
 *   0: 1 = cached (temporal) val/col/y streams in SpMV, 2 = streamed also where the default is cached (the irregular product tile)
 *              1: 1 = narrow loads in the CSR kernel
 *   2: workgroup map: 0 = operator's choice, < 0 identity, 1 = contiguous range per XCD, P >= 8 = strips of P
 *   3: 1 = hipStreamSynchronize instead of the event spin wait    4: long-row threshold (> 0), < 0 = no split
 *   5: 1 = unfused MGS chain, 2 = launch-lean MGS without graphs, 3 = one hipGraph per GMRES column, 4 = single-launch MGS on all XCDs (not the XCD-local form)
 *   7: cache hints of the CG vector kernels
 *   8: 1 = CSR row-block layout only (read at mik_csr_create and at launch)    9: 1 = nothing enqueued ahead of the host (GMRES: the next Arnoldi column; CG: the head of the next step)
 *  10: 1 = scalar results through hipMemcpyAsync + event spin instead of the publish kernel + mailbox spin
 *  11: 1 = no slice-constant values        12: 1 = no per-slice-offset layout
 *  13: bit mask switching the Krylov-basis streaming hints off    14: CSR kernel: 0 = by operator, 1 = register-staged products, 2 = LDS-DMA tile + per-row gather
 *  15: long-row segment length (> 0; read at mik_csr_create)   16: slices per workgroup of the layout-5 kernels (1 / 2 / 4)
 *  17: 1 = layout 5 through flat loads (k_spmv_sdiac)              18: 1 = k_spmv_sdiab without the compiled-in slot class
 *  19: 1 = layout 5 with one row per lane (k_spmv_sdiab instead of k_spmv_sdiab2)
 *  20: 1 = mik_csr_create on the host path only, 2 = device transpose but the host builders of layouts 6 / 1                 21: DGKS rounds inside the single-launch kernel (1..3; read at mik_gmres_create)
 *  22: 1 = PCG with a diagonal Pl as three vector sweeps (c = Pl \\ r and rho apart) instead of two (read at mik_cg_create)
 *  23: 1 = CG updates x in the step's own sweep (read at mik_cg_create)      24: 1 = the halo of a row-partitioned CG step after the whole sweep over u
 *  25: 1 = mik_bicgstab_step / mik_minres_step with separate finaliser launches at every size (the form used beyond 1,024 reduction segments);
 *      2 = no SpMV epilogues (the Lanczos step of MINRES, sigma / rho of BiCGStab(l) as sweeps of their own in the vector shape; no rho kept from the MR sweep)
 *  26: 1 = row-partitioned CG step with the separate alpha launch (k_cgd_alpha) instead of alpha formed inside the update sweep
 *  31: 1 = GMRES without the single-launch Gram-Schmidt kernels (read at mik_gmres_create)
 *  30: 1 = treat the next single-launch Gram-Schmidt column as timed out (exercises the fall-back to the multi-launch chains)
 *  28: jagged slices (layout 1): 1 = never, 2 = whenever the operator has no structured layout (read at mik_csr_create)
 *   6: transports of the row-partitioned CG (bits): 1 = the side stream ordered by events instead of mailbox flags, 2 = the two scalars of a step over
 *      RCCL although a mailbox is connected, 4 = ... through the mailbox even in a world of one, 8 = ... through the one-wave gather launches
 *      (k_mail_gather) instead of inside the finalising kernels
 *  29: x windows in LDS for the product-tile CSR kernel (k_spmv_rowblock XWIN): 1 = never built (read at mik_csr_create), 2 = built but not used at launch
 *  27: direction of the streaming launches of a plain CG step (bit 0 / 1 / 2: the u sweep / the SpMV / the update walk from
 *      the end of the vectors; 8: every launch against the one before it) -- no effect at the default cache hints */
int mik_set_tuning(int key, int value);
int mik_ctx_set_tuning(mik_ctx *ctx, int key, int value);
/* Host-only: the rule that places a 256-row block's window of x in LDS (k_spmv_rowblock XWIN) on caller-supplied per-block statistics
 * (first / last referenced column, entry count).  win_lo[b] = first column of block b's window (16-byte aligned) or -1 (the block gathers from
 * memory); *span = common window length in elements, 0 = no window table.  Guarantee the tests check: win_lo[b] >= 0 implies
 * win_lo[b] <= first_col[b], last_col[b] < win_lo[b] + span <= n_cols. */
int mik_dev_xwin_plan(int64_t n_blocks, const int *first_col, const int *last_col, const int *entries, int elem_size, int64_t n_cols,
                      int64_t total_entries, int *win_lo, int *span);
#ifdef __cplusplus
}
#endif
#endif /* MIK_DEV_H */
