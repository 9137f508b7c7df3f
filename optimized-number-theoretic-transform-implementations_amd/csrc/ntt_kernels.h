/*
 * ntt_kernels.h -- gfx950 kernels built from the templates of ntt_core.h.
 *
 * fused_kernel : one workgroup transforms whole 2^LOGN blocks.  16 coefficients
 *                per thread live in VGPRs for up to four stages at a time; the
 *                block crosses LDS once per stage group and HBM exactly twice
 *                (one coalesced read, one coalesced write): 16 bytes of HBM
 *                traffic per coefficient per transform, the algorithmic minimum
 *                (SURVEY 8d).  No MFMA: 53/64-bit modular butterflies are
 *                element-wise VALU work.
 * column_kernel: strided passes for the leading stages of N > 2^14 and for tiny N.
 * fused_product_kernel: c = a * b in Z_q[X]/(X^N+1) with b never leaving the CU: forward transform of b, product
 *                with a^ in registers, inverse transform (N = 2^14 whole polynomials; block by block for
 *                N = 2^15..2^17); the inverse half reads the forward LDS twiddle table in mirrored order.
 * twophase_kernel: both passes of a 2^16 / 2^17 transform inside one workgroup (optional; fabric-bound, see
 *                DESIGN.md section 3).
 * team_kernel  : both passes of a 2^15 .. 2^17 transform as ITEMS of one persistent launch: per-XCD in-order queues
 *                (the XCD is read from HW_REG_XCC_ID), per-polynomial hand-off counters, the intermediate kept in the
 *                XCD's L2 / the Infinity Cache; team_product_kernel: the same scheme with three item kinds for a whole
 *                product (column stages of both operands, block products -- both blocks through their block stages,
 *                product, inverse block stages --, inverse column stages; or, a^ given, the b-chain only).
 * MULTI        : kernel variants that serve several RNS limbs in one launch (a LimbRec per limb in the kernel arguments).
 * The same kernels serve four arithmetic policies (ntt_arith.h): FP64 with a reduction schedule, FP64 for moduli up
 * to 2^52, the reference's integer radix-2 arithmetic and its radix-4 formulation.
 *
 * Launch geometry (wave64, 256 CUs): LOGN=14 -> one persistent 1024-thread
 * workgroup per CU (16 waves, 4 per SIMD, <=128 VGPRs, no scratch) using 158 KiB
 * of the CU's 160 KiB LDS (128.1 KiB exchange buffer + 30 KiB twiddle table);
 * 2^13 -> one 512-thread workgroup (128 KiB: every per-lane twiddle in LDS),
 * 2^12 -> four 256-thread workgroups (39.6 KiB each).  Blocks >= 2^12 run the
 * persistent loops below (register prefetch of the next block); smaller blocks
 * pack several per 256-thread workgroup, share LDS twiddle tables from 2^8 up and
 * rely on multiple resident workgroups instead of a prefetch.
 * Tuning history and rejected variants: profiles/r01/ablations.txt, profiles/r02/ablations.txt.
 */
#pragma once
#include <hip/hip_runtime.h>

#include "ntt_core.h"
#include "ntt_passplan.h"

/* the kernels, by concern (round 6: one 3,800-line header before) */
#include "ntt_kernels_block.h"
#include "ntt_kernels_team.h"
#include "ntt_kernels_products.h"
#include "ntt_kernels_launch.h"
