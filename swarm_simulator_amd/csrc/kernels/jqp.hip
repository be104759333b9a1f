// jqp.hip — the JOINT RBP QP (plan/sequential = false, the reference's code default: param.hpp:67, setBatch rbp_planner.hpp:857-859
// "batch_size = N, batch_iter = 1") spread over the whole chip: one QP over all N agents of a mission, solved by MANY workgroups.
//
// kernels/qp.hip runs a batch QP on ONE workgroup; that is the right shape for the reference's sequential schedule (batches of 4..8
// agents, thousands of missions in flight) and the wrong one for the joint QP, whose Newton matrix has knot blocks of order
// nk = 9 N (576 for 64 agents, 2304 for 256).  Same mathematics here (null-space coordinates u_j per knot, Mehrotra predictor-
// corrector with the wide-neighbourhood step rule and the 1e-9 dual regularisation, see qp.hip), different machine mapping:
//
//   * a LAUNCH PER PHASE instead of a barrier per phase: every phase of an interior-point iteration is a kernel over all missions
//     of the session (grid.y / grid.z = mission), the kernel boundary is the grid-wide barrier (~1.5 us), and the per-mission scalars
//     (mu, sigma, alpha, state) live in a small device record written by one-workgroup "control" kernels -- the host only polls
//     "is any mission still running" once per iteration;
//   * ROW SWEEPS over the whole chip: a thread owns one control point of one agent and a CHUNK of 16 partner agents (N / 16 chunks
//     per control point give N * ceil(N / 16) workgroups per mission); a pair row is computed by both its agents from the single
//     stored (s, z) -- bit-identical arithmetic in canonical orientation -- and written back by the lower agent only; the partial
//     3x3 accumulators of the chunks are summed in a fixed order by their consumers (no floating-point atomics: bit-reproducible);
//   * the BLOCK-TRIDIAGONAL FACTORISATION keeps explicit inverses of the knots' Schur complements, S_j^-1, computed by a BLOCKED
//     SYMMETRIC SWEEP (Gauss-Jordan without pivoting on an SPD matrix) in 64 x 64 tiles on v_mfma_f64_16x16x4_f64:  step k inverts
//     the pivot tile, forms the panel Y_I = B_Ik P (one launch), and applies the rank-64 update B_IJ -= Y_I B_Jk' to every tile of
//     the lower triangle (one launch) -- uniform parallelism (nblk (nblk + 1) / 2 tiles) in every step, no triangular solves, one
//     kernel boundary per dependent step, nk^3 flops per knot (the same as Cholesky + triangular inverse).  The pivot tile of step
//     k + 1 is inverted by the workgroup that has just updated it (look-ahead: off the other tiles' path).  Because the coupling
//     blocks T_{j+1,j} are 3x3-block diagonal, the Schur update T_{j+1} - C S_j^-1 C' is an O(nk^2) elementwise pass fused with the
//     assembly of T_{j+1} from the sweeps' accumulators (T_j is never stored).  The elimination is TWISTED (both ends towards the
//     middle knot): half the dependent depth, the two chains share every launch;
//   * SUBSTITUTIONS are matrix-vector products with the stored inverses (16-row slabs, one launch per knot step and direction).
//   * two SCHEDULES of that sweep: look-ahead as described (few missions: the dependent chain is what counts) and bulk (eight or more
//     resident missions: jq_update_bulk without LDS at three workgroups per CU, the pivot inverse in a launch of its own);
//   * on top of Mehrotra's direction ONE centrality corrector per iteration (Gondzio) on the factorisation already paid for, and a
//     SAFEGUARD that takes back a step which loses the dual residual after an acceptable iterate (explicit inverses at Newton weights
//     of 1e9: see jq_ctrl(1));
//   * the active-set POLISH (jqp_polish.inc) reuses the tile sweep for S_AA^-1, rank-revealing there (dependent active rows are deleted
//     from a dual solve instead of eliminated).
//
// Layout: knot matrices are TILE-MAJOR (64 x 64 tiles of 32 KB, row-major inside), order nkp = nk rounded up to 64 (identity
// padding).  During the sweep only tiles I >= J are valid; the last step writes S_j^-1 with both triangles.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "jqp.h"

#pragma clang fp contract(fast)

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int JT = 64;          // tile order
constexpr int JTT = JT * JT;    // doubles per tile
constexpr int JQ_MAX_ITERS = 250;
constexpr double JQ_MU0 = 3e-1, JQ_SFLOOR = 1e-1, JQ_DREG = 1e-9, JQ_STEP_FRAC = 0.997, JQ_NBHD_GAMMA = 1e-3;

// per-mission state record (doubles)
enum { ST_STATE = 0 /* 0 running, 1 converged, 2 failed */, ST_ITER, ST_PAR /* which (s, z) pair is current */, ST_RETRY, ST_MU, ST_GAP, ST_PRES,
       ST_DRES, ST_SIGMU, ST_ALPHA, ST_APPLIED, ST_BT, ST_NROWS, ST_KKT, ST_FLOPS, ST_REASON, ST_AAFF, ST_POLISHED,
       ST_GO /* polish requested for this mission (set by the control kernel, cleared by the polish) */, ST_FINAL /* ... after convergence */,
       ST_TRIES /* early polish attempts so far */, ST_NREF /* refinement steps per Newton solve in this iteration */, ST_BADPIV, ST_DREG /* dual regularisation of this iteration's Newton system */, ST_DREGN /* ... of the next */, ST_PSTATE /* polish: see jqp_polish.inc */, ST_RDONE,
       ST_GACT /* centrality corrector: this mission tries one */, ST_GOK /* ... and took it: dx and the rows' target shifts T are in force */,
       ST_ATR /* trial step length */, ST_AP /* step length of the Mehrotra direction */,
       ST_PACC /* the iterate before the last step was acceptable (pres < 1e-9, dres < 1e-7, mu < 5e-8) */, ST_PMU, ST_PPRES, ST_PDRES,
       ST_REVERT /* the last step is being taken back (jq_unstep) */,
       ST_PTAU /* polish: this mission's deletion threshold is pol_tau * ST_PTAU (1, tightened by jp_check when a deleted row does not verify) */,
       ST_CONT /* continuations: final polish attempts refused so far, each followed by another decade of the interior-point method */,
       ST_N = 40 };
constexpr int JQ_MAX_CONT = 3;  // the exit threshold of a mission is exit_mu * 10^-ST_CONT: 1e-9, then 1e-10, 1e-11, 1e-12
// reduction slots (each [4 components][nred workgroups])
enum { RS_BUILD = 0 /* sum0 = gap, vmax = pres */, RS_POST /* dmax, gmax */, RS_AFF /* vmax, sum0, sum1, sum2 */, RS_STEP /* vmax */,
       RS_UP /* vmin */, RS_INIT /* pinned-row violation */, RS_VERIFY /* polish: worst violation at the trial point */, RS_NSLOT };

struct JDims {
    int N, M, oq, nj, nk, nkp, nblk, npair, nch, ncp;
};
__host__ __device__ inline JDims jdims(int N, int M) {
    JDims d;
    d.N = N, d.M = M, d.oq = 6 * M, d.nj = M - 1, d.nk = 9 * N, d.nkp = (d.nk + JT - 1) / JT * JT, d.nblk = d.nkp / JT;
    d.npair = N * (N - 1) / 2, d.nch = (N + JQ_PC - 1) / JQ_PC, d.ncp = N * d.oq;
    return d;
}

__constant__ double jc_Qbase[36] = {720,  -1800, 1200,  0,     0,     -120, -1800, 4800,  -3600, 0,     600,   0,
                                    1200, -3600, 3600,  -1200, 0,     0,    0,     0,     -1200, 3600,  -3600, 1200,
                                    0,    600,   0,     -3600, 4800,  -1800, -120, 0,     0,     1200,  -1800, 720};

__device__ inline size_t pair_index(int N, int qi, int qj) { return (size_t)qi * N - (size_t)qi * (qi + 1) / 2 + (qj - qi - 1); }

// element (r, c) of a tile-major matrix with nblk tile columns
__device__ __forceinline__ size_t telem(int nblk, int r, int c) { return ((size_t)(r >> 6) * nblk + (c >> 6)) * JTT + (size_t)(r & 63) * JT + (c & 63); }

__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// ---- deterministic block reduction (256 threads): op 0 sum, 1 max, 2 min ------------------------------------------------
__device__ __forceinline__ double red_op(double a, double b, int op) { return op == 0 ? a + b : (op == 1 ? fmax(a, b) : fmin(a, b)); }
__device__ inline double block_reduce(double v, int op, double* red /* >= 4 doubles of LDS */) {
    for (int o = 32; o > 0; o >>= 1) v = red_op(v, __shfl_xor(v, o), op);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double r = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = red_op(r, red[i], op);
    return r;
}

struct Ws {  // pointers into one mission's workspace
    double *st, *segsc, *Lk, *Dk, *Ek, *boxlo, *boxhi;
    double *bs[2], *bz[2], *ps[2], *pz[2], *pwgt, *acc, *dxa, *dx, *rbase, *rhs, *wv, *red, *Y, *P, *scr, *inv, *rhsc, *dx2, *tb, *tp;
};
__device__ __forceinline__ Ws carve(const JArgs& A, int mission) {
    double* b = A.ws + (size_t)mission * A.L.stride;
    const JLayout& L = A.L;
    Ws w;
    w.st = b + L.o_state, w.segsc = b + L.o_segsc, w.Lk = b + L.o_Lk, w.Dk = b + L.o_Dk, w.Ek = b + L.o_Ek;
    w.boxlo = b + L.o_boxlo, w.boxhi = b + L.o_boxhi;
    for (int p = 0; p < 2; ++p) w.bs[p] = b + L.o_bs[p], w.bz[p] = b + L.o_bz[p], w.ps[p] = b + L.o_ps[p], w.pz[p] = b + L.o_pz[p];
    w.pwgt = b + L.o_pwgt, w.acc = b + L.o_acc, w.dxa = b + L.o_dxa, w.dx = b + L.o_dx, w.rbase = b + L.o_rbase, w.rhs = b + L.o_rhs;
    w.wv = b + L.o_wv, w.red = b + L.o_red, w.Y = b + L.o_Y, w.P = b + L.o_P, w.scr = b + L.o_scr, w.inv = b + L.o_inv;
    w.rhsc = b + L.o_rhsc, w.dx2 = b + L.o_dx2, w.tb = b + L.o_tb, w.tp = b + L.o_tp;
    return w;
}
__device__ __forceinline__ double* red_slot(const Ws& w, const JLayout& L, int slot, int comp) { return w.red + ((size_t)slot * 4 + comp) * L.nred; }

#define JQ_POLISH_PART 1
#include "jqp_polish.inc"
#undef JQ_POLISH_PART

// ------------------------------------------------------------------------------------------------------------------------
// setup: mission constants (qp.hip mission_constants), SFC box per (agent, segment) (rbp_planner.hpp:447-453), pinned end control
// points (rows 0-5 of Aeq_base, :380-387), state record.  One workgroup per mission.
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void jq_setup(JArgs A) {
    const DevSession& S = A.S;
    const int mission = blockIdx.x, tid = threadIdx.x;
    const Ws w = carve(A, mission);
    const int N = S.N, M = S.Mk[mission], MS = S.M;
    const JDims d = jdims(N, M);
    const double* T = S.T + (size_t)mission * (MS + 1);
    if (tid == 0) {
        for (int i = 0; i < ST_N; ++i) w.st[i] = 0.0;
        w.st[ST_STATE] = S.status[mission] != 0 ? 2.0 : 0.0;
        w.st[ST_NROWS] = (double)((size_t)(d.oq - 6) * (6 * (size_t)N + d.npair));
        w.st[ST_DREG] = w.st[ST_DREGN] = A.dreg_mode == 0 ? JQ_DREG : fmin(A.dreg_max, fmax(JQ_DREG, A.dreg_scale * JQ_MU0));
    }
    __syncthreads();
    for (int m = tid; m < M; m += 256) w.segsc[m] = pow(T[m + 1] - T[m], -5.0);
    for (int j = tid; j <= M; j += 256) {
        double* L = w.Lk + 9 * j;
        for (int e = 0; e < 9; ++e) L[e] = 0;
        if (j >= 1 && j < M) {
            const double r = (T[j] - T[j - 1]) / (T[j + 1] - T[j]);
            L[0] = (1 + r) * (1 + r), L[1] = -2 * r * (1 + r), L[2] = r * r;
            L[3] = 1 + r, L[4] = -r, L[5] = 0;
            L[6] = 1, L[7] = 0, L[8] = 0;
        }
    }
    __syncthreads();
    for (int j = tid; j <= M; j += 256) {
        double* D = w.Dk + 9 * j;
        double* E = w.Ek + 9 * j;
        for (int e = 0; e < 9; ++e) D[e] = 0, E[e] = 0;
        if (j >= 1 && j < M) {
            const double sl = pow(T[j] - T[j - 1], -5.0), sr = pow(T[j + 1] - T[j], -5.0);
            const double* L = w.Lk + 9 * j;
            double QL[9];
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) {
                    double s = 0;
                    for (int c = 0; c < 3; ++c) s += jc_Qbase[6 * (3 + a) + 3 + c] * sl * L[3 * c + b];
                    QL[3 * a + b] = s;
                }
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) {
                    double s = 0;
                    for (int c = 0; c < 3; ++c) s += L[3 * c + a] * QL[3 * c + b];
                    D[3 * a + b] = 2 * (s + jc_Qbase[6 * a + b] * sr);
                }
            if (j + 1 < M) {
                const double* Ln = w.Lk + 9 * (j + 1);
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) {
                        double s = 0;
                        for (int c = 0; c < 3; ++c) s += jc_Qbase[6 * a + 3 + c] * sr * Ln[3 * c + b];
                        E[3 * a + b] = 2 * s;
                    }
            }
        }
    }
    // SFC box of every (agent, segment): first box with end time >= T[m+1]
    for (int a = tid; a < N; a += 256) {
        int nbx = S.sfc_count[(size_t)mission * N + a];
        if (nbx <= 0) {
            atomicCAS(&S.status[mission], 0, (int)RBP_ERR_BAD_ARGUMENT);
            w.st[ST_STATE] = 2.0;
            nbx = 1;
        }
        const double* bt = S.sfc_time + ((size_t)mission * N + a) * S.max_boxes;
        int bi = 0;
        for (int m = 0; m < M; ++m) {
            while (bi < nbx && bt[bi] < T[m + 1]) bi++;
            const int sel = bi < nbx ? bi : nbx - 1;
            const double* bx = S.sfc_box + (((size_t)mission * N + a) * S.max_boxes + sel) * 6;
            for (int k = 0; k < 3; ++k) w.boxlo[((size_t)a * M + m) * 3 + k] = bx[k], w.boxhi[((size_t)a * M + m) * 3 + k] = bx[3 + k];
        }
    }
    double* ctrl = S.ctrl + (size_t)mission * N * 3 * 6 * MS;
    for (int it = tid; it < N * 3; it += 256) {
        const int a = it / 3, k = it % 3;
        const double* stt = S.start + ((size_t)mission * N + a) * 9;
        const double* gl = S.goal + ((size_t)mission * N + a) * 9;
        const double h0 = T[1] - T[0], hT = T[M] - T[M - 1];
        double* x = ctrl + ((size_t)a * 3 + k) * d.oq;
        x[0] = stt[k], x[1] = x[0] + h0 * stt[k + 3] / 5, x[2] = 2 * x[1] - x[0] + h0 * h0 * stt[k + 6] / 20;
        double* xe = x + 6 * (M - 1);
        xe[5] = gl[k], xe[4] = xe[5] - hT * gl[k + 3] / 5, xe[3] = 2 * xe[4] - xe[5] + hT * hT * gl[k + 6] / 20;
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// row sweeps.  Rows (G x <= h form, as in qp.hip and the oracle):
//   bound (a, k, side, j6):  +x <= hi / -x <= -lo                    rbp_planner.hpp:626-635
//   pair  (lo < hi, j6):     n . x_lo - n . x_hi <= -(r_lo + r_hi)   :668-679
// Control points j6 < 3 and j6 >= 6M - 3 are pinned: their rows are constants, checked once (PASS_INIT) against 1e-6.
// ------------------------------------------------------------------------------------------------------------------------
enum { PASS_INIT = 0, PASS_BUILD, PASS_AFF, PASS_STEP, PASS_UPBUILD, PASS_CAND, PASS_VERIFY,
       PASS_KMUL_A /* J'W J dxa for the iterative refinement of a Newton solve */, PASS_KMUL_D /* ... of dx */, PASS_KMUL_G /* ... of dx2 */,
       PASS_GOND /* centrality corrector: target shifts T of the rows and the change of the right-hand side */,
       PASS_STEPG /* step length of the trial direction dx2 with T */ };

struct PassIO {
    double sigma_mu, alpha, dreg, dregn, mu0, sfloor, atr, mut;
    double sum0, sum1, sum2, vmax, vmin;
};

// one row: see row_op in qp.hip (same arithmetic).  s, z: current state; out: new state (INIT, UPBUILD) through so / zo2.
template <int PASS>
__device__ __forceinline__ void row_op(double slack, double ga, double gd, double s, double z, PassIO& io, double cw, double& wgt, double& v,
                                       double& zo, double& sn_out, double& zn_out, double tt = 0.0) {
    if (PASS == PASS_INIT) {
        const double s0 = slack < io.sfloor ? io.sfloor : slack;
        sn_out = s0, zn_out = io.mu0 / s0;
    } else if (PASS == PASS_BUILD) {
        const double rg = s - slack;
        wgt = z * fast_rcp(s + io.dregn * z);
        v = -wgt * (rg - s);
        zo = z;
        io.sum0 += cw * s * z;
        io.vmax = fmax(io.vmax, fabs(rg));
    } else if (PASS == PASS_AFF) {
        const double rg = s - slack;
        const double iz = fast_rcp(z), is = fast_rcp(s);
        wgt = z * fast_rcp(s + io.dreg * z);
        const double dza = wgt * (ga + rg - s);
        const double dsa = -s - s * dza * iz;
        const double cc = dsa * dza;
        io.vmax = fmax(io.vmax, fmax(-dsa * is, -dza * iz));
        io.sum0 += cw * s * z, io.sum1 += cw * (s * dza + z * dsa), io.sum2 += cw * cc;
        v = -wgt * (rg - s - cc * iz);
        wgt = wgt * iz;
    } else if (PASS == PASS_STEP || PASS == PASS_STEPG) {
        const double rg = s - slack;
        const double iz = fast_rcp(z), is = fast_rcp(s);
        wgt = z * fast_rcp(s + io.dreg * z);
        const double dza = wgt * (ga + rg - s);
        const double cc = (-s - s * dza * iz) * dza;
        const double rcc = s * z + cc - io.sigma_mu - tt;
        const double dz = wgt * (gd + rg - rcc * iz);
        const double ds = -(rcc + s * dz) * iz;
        io.vmax = fmax(io.vmax, fmax(-ds * is, -dz * iz));
    } else if (PASS == PASS_GOND) {
        // Gondzio's centrality corrector: at the trial step length the complementarity products of the Mehrotra direction are projected
        // onto [0.1, 10] x (sigma mu); the shift t = projected - actual (not below -10 sigma mu) moves the row's target, and the
        // direction is solved for again: rcc - t instead of rcc, i.e. the right-hand side changes by -G'(W t / z)
        const double rg = s - slack;
        const double iz = fast_rcp(z);
        const double w0 = z * fast_rcp(s + io.dreg * z);
        const double dza = w0 * (ga + rg - s);
        const double cc = (-s - s * dza * iz) * dza;
        const double rcc = s * z + cc - io.sigma_mu;
        const double dz = w0 * (gd + rg - rcc * iz);
        const double ds = -(rcc + s * dz) * iz;
        const double pr = (s + io.atr * ds) * (z + io.atr * dz);
        const double lo = 0.1 * io.mut, hi = 10.0 * io.mut;
        double t = (pr < lo ? lo : (pr > hi ? hi : pr)) - pr;
        t = fmax(t, -hi);
        sn_out = t;
        v = -w0 * t * iz;
    } else if (PASS == PASS_UPBUILD) {
        const double rg = s - (slack + io.alpha * gd);  // the old point: slack_old = slack + alpha * gd
        const double iz = fast_rcp(z);
        const double w0 = z * fast_rcp(s + io.dreg * z);
        const double dza = w0 * (ga + rg - s);
        const double cc = (-s - s * dza * iz) * dza;
        const double rcc = s * z + cc - io.sigma_mu - tt;
        const double dz = w0 * (gd + rg - rcc * iz);
        const double ds = -(rcc + s * dz) * iz;
        const double sn = s + io.alpha * ds, zn = z + io.alpha * dz;
        sn_out = sn, zn_out = zn;
        io.vmin = fmin(io.vmin, sn * zn);
        const double rgn = sn - slack;
        wgt = zn * fast_rcp(sn + io.dregn * zn);
        v = -wgt * (rgn - sn);
        zo = zn;
        io.sum0 += cw * sn * zn;
        io.vmax = fmax(io.vmax, fabs(rgn));
    } else if (PASS == PASS_KMUL_A || PASS == PASS_KMUL_D || PASS == PASS_KMUL_G) {
        wgt = z * fast_rcp(s + io.dreg * z);  // the weight this iteration's Newton matrix was assembled with
        v = wgt * gd;
    }
}

template <int PASS>
__global__ __launch_bounds__(256) void jq_sweep(JArgs A) {
    const DevSession& S = A.S;
    const int mission = blockIdx.y, tid = threadIdx.x;
    const Ws w = carve(A, mission);
    __shared__ double red[8];
    if (w.st[ST_STATE] != 0.0) return;
    if (PASS == PASS_CAND || PASS == PASS_VERIFY) {
        if (w.st[ST_GO] == 0.0 || w.st[ST_PSTATE] != (double)(PASS == PASS_CAND ? PS_SOLVE : PS_PRIMAL)) return;
    } else if (w.st[ST_GO] != 0.0 || (PASS != PASS_UPBUILD && w.st[ST_RETRY] != 0.0))
        return;  // (a mission whose polish request is waiting for company is frozen: see launch_planner_joint)
    if (PASS == PASS_UPBUILD && A.retry_only && w.st[ST_RETRY] == 0.0) return;  // (a repeat launch: only for missions whose step was refused)
    constexpr bool kmulr = (PASS == PASS_KMUL_A || PASS == PASS_KMUL_D || PASS == PASS_KMUL_G);
    constexpr bool gond = (PASS == PASS_GOND);
    constexpr bool kmul = kmulr || gond;  // (three accumulators per control point: G' v)
    constexpr bool use_dx2 = (PASS == PASS_KMUL_G || PASS == PASS_STEPG);
    if (kmulr && w.st[ST_NREF] <= (double)A.ref_step) return;
    if ((gond || use_dx2 || A.gond_only) && w.st[ST_GACT] == 0.0) return;
    const bool use_t = PASS == PASS_STEPG || (PASS == PASS_UPBUILD && w.st[ST_GOK] != 0.0);
    const int N = S.N, M = S.Mk[mission], MS = S.M;
    const JDims d = jdims(N, M);
    const int oq = d.oq, ncp = d.ncp;
    const int a = blockIdx.x / d.nch, ch = blockIdx.x % d.nch;
    const Pol pol = pol_carve(A, mission);
    const int par = (int)w.st[ST_PAR];
    const double *bs = w.bs[par], *bz = w.bz[par], *ps = w.ps[par], *pz = w.pz[par];
    double *bs2 = w.bs[par ^ 1], *bz2 = w.bz[par ^ 1], *ps2 = w.ps[par ^ 1], *pz2 = w.pz[par ^ 1];
    if (PASS == PASS_INIT) bs2 = w.bs[par], bz2 = w.bz[par], ps2 = w.ps[par], pz2 = w.pz[par];
    const double* ctrl = S.ctrl + (size_t)mission * N * 3 * 6 * MS;
    const float* normals = S.rsfc_normal + (size_t)mission * S.npair * MS * 3;
    const double* radius = S.radius + (size_t)mission * N;
    constexpr bool build = (PASS == PASS_BUILD || PASS == PASS_UPBUILD);
    constexpr bool aff = (PASS == PASS_AFF);
    constexpr bool accum = build || aff || kmul;
    constexpr bool need_da = (PASS == PASS_AFF || PASS == PASS_STEP || PASS == PASS_STEPG || gond || PASS == PASS_UPBUILD);
    constexpr bool need_dd = (PASS == PASS_STEP || PASS == PASS_STEPG || PASS == PASS_UPBUILD || PASS == PASS_VERIFY || kmul);
    constexpr bool rd_sz = PASS != PASS_INIT && PASS != PASS_VERIFY;
    constexpr bool polish = (PASS == PASS_CAND || PASS == PASS_VERIFY);
    constexpr bool wr_sz = (PASS == PASS_INIT || PASS == PASS_UPBUILD);
    PassIO io;
    io.sigma_mu = w.st[ST_SIGMU], io.alpha = w.st[ST_ALPHA], io.dreg = w.st[ST_DREG], io.dregn = w.st[ST_DREGN];
    io.mu0 = A.tune[0], io.sfloor = A.tune[1];
    io.atr = w.st[ST_ATR], io.mut = fmax(w.st[ST_SIGMU], 1e-3 * w.st[ST_MU]);
    const double* dvec = PASS == PASS_KMUL_A ? w.dxa : (use_dx2 ? w.dx2 : w.dx);
    io.sum0 = io.sum1 = io.sum2 = 0, io.vmax = 0, io.vmin = 1e300;
    double pin_viol = 0;
    const double ra = radius[a];
    const int b0 = ch * JQ_PC, b1 = min(N, b0 + JQ_PC);
    for (int j6 = tid; j6 < oq; j6 += 256) {
        const bool pinned = (j6 < 3 || j6 >= oq - 3);
        if (pinned && PASS != PASS_INIT) continue;
        const int seg = j6 / 6;
        double xa[3], da[3], dd[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            xa[k] = ctrl[((size_t)a * 3 + k) * oq + j6];
            da[k] = need_da ? w.dxa[((size_t)a * 3 + k) * oq + j6] : 0.0;
            dd[k] = need_dd ? dvec[((size_t)a * 3 + k) * oq + j6] : 0.0;
        }
        double Sm[6] = {0, 0, 0, 0, 0, 0}, yv[3] = {0, 0, 0}, gz[3] = {0, 0, 0};
        if (ch == 0) {  // bound rows
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double hi = w.boxhi[((size_t)a * M + seg) * 3 + k], lo = w.boxlo[((size_t)a * M + seg) * 3 + k];
#pragma unroll
                for (int side = 0; side < 2; ++side) {
                    const size_t r = (size_t)(2 * k + side) * ncp + (size_t)a * oq + j6;
                    const double sg = side == 0 ? 1.0 : -1.0;
                    const double slack = side == 0 ? hi - xa[k] : xa[k] - lo;
                    if (pinned) {
                        pin_viol = fmax(pin_viol, -slack);
                        continue;
                    }
                    double wgt = 0, v = 0, zo = 0, sn = 0, zn = 0;
                    const double s = rd_sz ? bs[r] : 0.0, z = rd_sz ? bz[r] : 0.0;
                    if (polish) {
                        const int snap = (int)(((size_t)a * 3 + k) * oq + j6);
                        if (PASS == PASS_CAND) {
                            if (z > s || s < 1e-6)
                                pol_emit(pol, (int)r, a, -1, j6, k == 0 ? sg : 0.0, k == 1 ? sg : 0.0, k == 2 ? sg : 0.0, slack, snap, side == 0 ? hi : lo,
                                         fmax(z / s, 1e-300));
                        } else {
                            const double snv = slack - sg * dd[k];  // slack at x + dx
                            io.vmax = fmax(io.vmax, -snv);
                            const int q = pol.pos[r];
                            if (q >= 0)
                                pol.e[q] = snv;
                            else if (snv < -1e-11)
                                pol_emit(pol, (int)r, a, -1, j6, k == 0 ? sg : 0.0, k == 1 ? sg : 0.0, k == 2 ? sg : 0.0, slack, snap, side == 0 ? hi : lo, 1.0);
                        }
                        continue;
                    }
                    row_op<PASS>(slack, sg * da[k], sg * dd[k], s, z, io, 1.0, wgt, v, zo, sn, zn, use_t ? w.tb[r] : 0.0);
                    if (wr_sz) bs2[r] = sn, bz2[r] = zn;
                    if (gond) w.tb[r] = sn;
                    if (accum) {
                        const int dg = k == 0 ? 0 : (k == 1 ? 3 : 5);
                        if (build) {
                            Sm[dg] += wgt, gz[k] += sg * zo, yv[k] += sg * v;
                        } else if (kmul) {
                            Sm[k] += sg * v;
                        } else {
                            Sm[k] += sg * v, Sm[3 + k] += sg * wgt;
                        }
                    }
                }
            }
        }
        for (int b = b0; b < b1; ++b) {
            if (b == a) continue;
            const bool a_lo = a < b;
            const size_t pi = pair_index(N, a_lo ? a : b, a_lo ? b : a);
            const size_t r = pi * oq + j6;
            const float* nv = normals + (pi * M + seg) * 3;
            const double n0 = nv[0], n1 = nv[1], n2 = nv[2];
            double xb[3], gab = 0, gdb = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) xb[k] = ctrl[((size_t)b * 3 + k) * oq + j6];
            const double e0 = a_lo ? xb[0] - xa[0] : xa[0] - xb[0], e1 = a_lo ? xb[1] - xa[1] : xa[1] - xb[1],
                         e2 = a_lo ? xb[2] - xa[2] : xa[2] - xb[2];
            const double slack = n0 * e0 + n1 * e1 + n2 * e2 - (a_lo ? ra + radius[b] : radius[b] + ra);
            if (pinned) {
                pin_viol = fmax(pin_viol, -slack);
                continue;
            }
            if (need_da) {
                const double f0 = w.dxa[((size_t)b * 3 + 0) * oq + j6], f1 = w.dxa[((size_t)b * 3 + 1) * oq + j6],
                             f2 = w.dxa[((size_t)b * 3 + 2) * oq + j6];
                gab = a_lo ? n0 * (da[0] - f0) + n1 * (da[1] - f1) + n2 * (da[2] - f2) : n0 * (f0 - da[0]) + n1 * (f1 - da[1]) + n2 * (f2 - da[2]);
            }
            if (need_dd) {
                const double* dxp = dvec;
                const double f0 = dxp[((size_t)b * 3 + 0) * oq + j6], f1 = dxp[((size_t)b * 3 + 1) * oq + j6], f2 = dxp[((size_t)b * 3 + 2) * oq + j6];
                gdb = a_lo ? n0 * (dd[0] - f0) + n1 * (dd[1] - f1) + n2 * (dd[2] - f2) : n0 * (f0 - dd[0]) + n1 * (f1 - dd[1]) + n2 * (f2 - dd[2]);
            }
            double wgt = 0, v = 0, zo = 0, sn = 0, zn = 0;
            const double s = rd_sz ? ps[r] : 0.0, z = rd_sz ? pz[r] : 0.0;
            if (polish) {
                if (!a_lo) continue;  // (one copy of a pair row is enough here)
                const int rowid = (int)(6 * (size_t)ncp + r);
                if (PASS == PASS_CAND) {
                    if (z > s || s < 1e-6) pol_emit(pol, rowid, a, b, j6, n0, n1, n2, slack, -1, 0.0, fmax(z / s, 1e-300));
                } else {
                    const double snv = slack - gdb;
                    io.vmax = fmax(io.vmax, -snv);
                    const int q = pol.pos[rowid];
                    if (q >= 0)
                        pol.e[q] = snv;
                    else if (snv < -1e-11)
                        pol_emit(pol, rowid, a, b, j6, n0, n1, n2, slack, -1, 0.0, 1.0);
                }
                continue;
            }
            row_op<PASS>(slack, gab, gdb, s, z, io, a_lo ? 1.0 : 0.0, wgt, v, zo, sn, zn, use_t ? w.tp[r] : 0.0);
            if (wr_sz && a_lo) ps2[r] = sn, pz2[r] = zn;
            if (gond && a_lo) w.tp[r] = sn;
            if (accum) {
                const double sg = a_lo ? 1.0 : -1.0;
                if (build) {
                    if (a_lo) w.pwgt[r] = wgt;
                    Sm[0] += wgt * n0 * n0, Sm[1] += wgt * n0 * n1, Sm[2] += wgt * n0 * n2;
                    Sm[3] += wgt * n1 * n1, Sm[4] += wgt * n1 * n2, Sm[5] += wgt * n2 * n2;
                    const double zz = sg * zo, vv = sg * v;
                    gz[0] += zz * n0, gz[1] += zz * n1, gz[2] += zz * n2;
                    yv[0] += vv * n0, yv[1] += vv * n1, yv[2] += vv * n2;
                } else if (kmul) {
                    const double vv = sg * v;
                    Sm[0] += vv * n0, Sm[1] += vv * n1, Sm[2] += vv * n2;
                } else {
                    const double vv = sg * v, ww = sg * wgt;
                    Sm[0] += vv * n0, Sm[1] += vv * n1, Sm[2] += vv * n2;
                    Sm[3] += ww * n0, Sm[4] += ww * n1, Sm[5] += ww * n2;
                }
            }
        }
        if (accum && !pinned) {
            double* acc = w.acc + (size_t)ch * 12 * ncp + (size_t)a * oq + j6;
#pragma unroll
            for (int e = 0; e < (kmul ? 3 : 6); ++e) acc[(size_t)e * ncp] = Sm[e];
            if (build) {
#pragma unroll
                for (int e = 0; e < 3; ++e) acc[(size_t)(6 + e) * ncp] = yv[e], acc[(size_t)(9 + e) * ncp] = gz[e];
            }
        }
    }
    // per-workgroup partials of the sweep's reductions (summed in a fixed order by the control kernels)
    const int wg = blockIdx.x;
    if (PASS == PASS_INIT) {
        const double v = block_reduce(pin_viol, 1, red);
        if (tid == 0) red_slot(w, A.L, RS_INIT, 0)[wg] = v;
    } else if (PASS == PASS_BUILD || PASS == PASS_UPBUILD) {
        const double g = block_reduce(io.sum0, 0, red), p = block_reduce(io.vmax, 1, red);
        if (tid == 0) red_slot(w, A.L, RS_BUILD, 0)[wg] = g, red_slot(w, A.L, RS_BUILD, 1)[wg] = p;
        if (PASS == PASS_UPBUILD) {
            const double m = block_reduce(io.vmin, 2, red);
            if (tid == 0) red_slot(w, A.L, RS_UP, 0)[wg] = m;
        }
    } else if (PASS == PASS_AFF) {
        const double vm = block_reduce(io.vmax, 1, red), q0 = block_reduce(io.sum0, 0, red), q1 = block_reduce(io.sum1, 0, red),
                     q2 = block_reduce(io.sum2, 0, red);
        if (tid == 0) {
            red_slot(w, A.L, RS_AFF, 0)[wg] = vm, red_slot(w, A.L, RS_AFF, 1)[wg] = q0;
            red_slot(w, A.L, RS_AFF, 2)[wg] = q1, red_slot(w, A.L, RS_AFF, 3)[wg] = q2;
        }
    } else if (PASS == PASS_STEP || PASS == PASS_STEPG) {
        const double vm = block_reduce(io.vmax, 1, red);
        if (tid == 0) red_slot(w, A.L, RS_STEP, 0)[wg] = vm;
    } else if (PASS == PASS_VERIFY) {
        const double vm = block_reduce(io.vmax, 1, red);
        if (tid == 0) red_slot(w, A.L, RS_VERIFY, 0)[wg] = vm;
    }
}

// sum over the chunks' partial accumulators of component e at control point (a, j6)
__device__ __forceinline__ double acc_sum(const Ws& w, const JDims& d, int e, size_t cp) {
    double s = 0;
    for (int c = 0; c < d.nch; ++c) s += w.acc[((size_t)c * 12 + e) * d.ncp + cp];
    return s;
}

// ------------------------------------------------------------------------------------------------------------------------
// reduced-space right-hand sides.  One thread per (knot j, agent a, dim k):
//   post<0>: rbase = -F'(2Qx + G'z), rhs = rbase + F'(G'v) (predictor), partials of max|rbase| and max|2Qx + G'z|
//   post<1>: rhs = rbase + F'(part1 - sigma mu * part2)   (corrector; both parts were accumulated by the AFF sweep); kept in rhsc
//   post<2>: rhs = rhsc + F'(G'v)   (centrality corrector: v = -W t / z from the GOND sweep)
// ------------------------------------------------------------------------------------------------------------------------
template <int CORR>
__global__ __launch_bounds__(256) void jq_post(JArgs A) {
    const DevSession& S = A.S;
    const int mission = blockIdx.y, tid = threadIdx.x;
    const Ws w = carve(A, mission);
    __shared__ double red[8];
    if (w.st[ST_STATE] != 0.0 || w.st[ST_RETRY] != 0.0 || w.st[ST_GO] != 0.0) return;
    if (CORR == 2 && w.st[ST_GACT] == 0.0) return;
    const int N = S.N, M = S.Mk[mission], MS = S.M;
    const JDims d = jdims(N, M);
    const int oq = d.oq, nu = 3 * N;
    const double* ctrl = S.ctrl + (size_t)mission * N * 3 * 6 * MS;
    const double sigma_mu = w.st[ST_SIGMU];
    double dmax = 0, gmax = 0;
    const int it = blockIdx.x * 256 + tid;
    if (it < d.nj * nu) {
        const int j = it / nu + 1, u = it % nu, a = u / 3, k = u % 3;
        const double* L = w.Lk + 9 * j;
        const size_t o0 = (size_t)(j - 1) * d.nkp + u * 3;
        double g[6], y[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int j6 = 6 * (j - 1) + 3 + q;
            const size_t cp = (size_t)a * oq + j6;
            if (CORR == 2) {
                y[q] = acc_sum(w, d, k, cp);
            } else if (CORR == 1) {
                y[q] = acc_sum(w, d, k, cp) - sigma_mu * acc_sum(w, d, 3 + k, cp);
            } else {
                const int m = j6 / 6, i = j6 % 6;
                const double* xs = ctrl + ((size_t)a * 3 + k) * oq + 6 * m;
                double gv = 0;
#pragma unroll
                for (int jj = 0; jj < 6; ++jj) gv += jc_Qbase[6 * i + jj] * xs[jj];
                gv *= 2 * w.segsc[m];
                gv += acc_sum(w, d, 9 + k, cp);
                g[q] = -gv;
                gmax = fmax(gmax, fabs(gv));
                y[q] = acc_sum(w, d, 6 + k, cp);
            }
        }
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            double rb;
            if (CORR == 2)
                rb = w.rhsc[o0 + e];
            else if (CORR == 1)
                rb = w.rbase[o0 + e];
            else {
                rb = g[3 + e] + L[0 + e] * g[0] + L[3 + e] * g[1] + L[6 + e] * g[2];
                w.rbase[o0 + e] = rb;
                dmax = fmax(dmax, fabs(rb));
            }
            const double r = rb + y[3 + e] + L[0 + e] * y[0] + L[3 + e] * y[1] + L[6 + e] * y[2];
            w.rhs[o0 + e] = r;
            if (CORR == 1) w.rhsc[o0 + e] = r;
        }
    }
    if (!CORR) {
        const double dm = block_reduce(dmax, 1, red), gm = block_reduce(gmax, 1, red);
        if (tid == 0) red_slot(w, A.L, RS_POST, 0)[blockIdx.x] = dm, red_slot(w, A.L, RS_POST, 1)[blockIdx.x] = gm;
    }
}

// dx[a][k][j6] = F du (du = the solution left in rhs); which: 0 -> dxa, 1 -> dx, 2 -> dx2
__global__ __launch_bounds__(256) void jq_apply_F(JArgs A, int which) {
    const DevSession& S = A.S;
    const int mission = blockIdx.y;
    const Ws w = carve(A, mission);
    if (w.st[ST_STATE] != 0.0 || w.st[ST_RETRY] != 0.0 || w.st[ST_GO] != 0.0) return;
    if (A.gond_only && w.st[ST_GACT] == 0.0) return;
    const int N = S.N, M = S.Mk[mission];
    const JDims d = jdims(N, M);
    const int nu = 3 * N, it = blockIdx.x * 256 + threadIdx.x;
    if (it >= d.nj * nu) return;
    const int j = it / nu + 1, u = it % nu;
    const double* uu = w.rhs + (size_t)(j - 1) * d.nkp + u * 3;
    const double* L = w.Lk + 9 * j;
    const double u0 = uu[0], u1 = uu[1], u2 = uu[2];
    double* o = (which == 2 ? w.dx2 : (which ? w.dx : w.dxa)) + (size_t)u * d.oq + 6 * (j - 1) + 3;
    o[0] = L[0] * u0 + L[1] * u1 + L[2] * u2;
    o[1] = L[3] * u0 + L[4] * u1 + L[5] * u2;
    o[2] = L[6] * u0 + L[7] * u1 + L[8] * u2;
    o[3] = u0, o[4] = u1, o[5] = u2;
}

// x += (alpha - applied) dx
__global__ __launch_bounds__(256) void jq_stepx(JArgs A) {
    const DevSession& S = A.S;
    const int mission = blockIdx.y;
    const Ws w = carve(A, mission);
    if (w.st[ST_STATE] != 0.0 || w.st[ST_GO] != 0.0 || (A.retry_only && w.st[ST_RETRY] == 0.0)) return;
    const int N = S.N, M = S.Mk[mission], MS = S.M;
    const int nx = N * 3 * 6 * M, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nx) return;
    double* ctrl = S.ctrl + (size_t)mission * N * 3 * 6 * MS;
    ctrl[i] += (w.st[ST_ALPHA] - w.st[ST_APPLIED]) * w.dx[i];
}

// the safeguard of jq_ctrl(1) takes the last step back: x -= alpha dx (dx is still the direction of that step)
__global__ __launch_bounds__(256) void jq_unstep(JArgs A) {
    const DevSession& S = A.S;
    const int mission = blockIdx.y;
    const Ws w = carve(A, mission);
    if (w.st[ST_REVERT] != 1.0) return;
    const int N = S.N, M = S.Mk[mission], MS = S.M;
    const int nx = N * 3 * 6 * M, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nx) return;
    double* ctrl = S.ctrl + (size_t)mission * N * 3 * 6 * MS;
    ctrl[i] -= w.st[ST_ALPHA] * w.dx[i];
}
__global__ void jq_unstep_done(JArgs A) {
    const Ws w = carve(A, blockIdx.x);
    if (threadIdx.x == 0 && w.st[ST_REVERT] == 1.0) w.st[ST_REVERT] = 2.0;  // (taken back once; the record keeps that it happened)
}

// the centrality corrector was accepted: its direction becomes THE direction (the rows' shifts T stay in tb / tp, flagged by ST_GOK)
__global__ __launch_bounds__(256) void jq_gcopy(JArgs A) {
    const DevSession& S = A.S;
    const int mission = blockIdx.y;
    const Ws w = carve(A, mission);
    if (w.st[ST_STATE] != 0.0 || w.st[ST_RETRY] != 0.0 || w.st[ST_GO] != 0.0 || w.st[ST_GOK] == 0.0) return;  // (ST_GOK lasts until the next iteration's step sweep)
    const int nx = S.N * 3 * 6 * S.Mk[mission], i = blockIdx.x * 256 + threadIdx.x;
    if (i < nx) w.dx[i] = w.dx2[i];
}

// ---- iterative refinement of a Newton solve.  The substitutions multiply with explicit inverses: their residual is cond(K) eps, not eps
// (the price of having no triangular solves), and with Newton weights of 1e9 the last interior-point iterations lose the dual residual
// without it.  K du is formed matrix-free: K0 F du from the jerk Gram matrices, J'W J F du by a row sweep (PASS_KMUL).
// op 0: rhs0 <- rhs (before the first solve)
// op 1: after the KMUL sweep of refinement step `rs`: dusave <- du (rs == 0) ;  rhs <- rhs0 - K du
// op 2: after the correction solve: rhs <- dusave + rhs ; dusave <- that
__global__ __launch_bounds__(256) void jq_refine(JArgs A, int op, int which) {
    const DevSession& S = A.S;
    const int mission = blockIdx.y;
    const Ws w = carve(A, mission);
    if (w.st[ST_STATE] != 0.0 || w.st[ST_RETRY] != 0.0 || w.st[ST_GO] != 0.0) return;
    if (w.st[ST_NREF] <= (double)(op == 0 ? 0 : A.ref_step)) return;
    if (A.gond_only && w.st[ST_GACT] == 0.0) return;
    const int N = S.N, M = S.Mk[mission];
    const JDims d = jdims(N, M);
    const int oq = d.oq, nu = 3 * N, it = blockIdx.x * 256 + threadIdx.x;
    if (it >= d.nj * nu) return;
    const int j = it / nu + 1, u = it % nu, a = u / 3, k = u % 3;
    const size_t o0 = (size_t)(j - 1) * d.nkp + u * 3;
    double* rhs0 = w.wv + (size_t)A.L.njS * A.L.nkpS;      // [nj * nkp] behind wv
    double* dusave = rhs0 + (size_t)A.L.njS * A.L.nkpS;
    if (op == 0) {
#pragma unroll
        for (int e = 0; e < 3; ++e) rhs0[o0 + e] = w.rhs[o0 + e];
        return;
    }
    if (op == 2) {
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const double v = dusave[o0 + e] + w.rhs[o0 + e];
            w.rhs[o0 + e] = v, dusave[o0 + e] = v;
        }
        return;
    }
    const double* dxp = (which == 2 ? w.dx2 : (which ? w.dx : w.dxa)) + ((size_t)a * 3 + k) * oq;
    const double* L = w.Lk + 9 * j;
    double g[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int j6 = 6 * (j - 1) + 3 + q, m = j6 / 6, i = j6 % 6;
        double gv = 0;
#pragma unroll
        for (int jj = 0; jj < 6; ++jj) gv += jc_Qbase[6 * i + jj] * dxp[6 * m + jj];
        g[q] = 2 * w.segsc[m] * gv + acc_sum(w, d, k, (size_t)a * oq + j6);
    }
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const double kdu = g[3 + e] + L[0 + e] * g[0] + L[3 + e] * g[1] + L[6 + e] * g[2];
        if (A.ref_step == 0) dusave[o0 + e] = w.rhs[o0 + e];
        w.rhs[o0 + e] = rhs0[o0 + e] - kdu;
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// control kernels: one workgroup per mission finishes the reductions and takes the decisions of the interior-point loop
// ------------------------------------------------------------------------------------------------------------------------
__device__ inline double red_final(const double* part, int n, int op, double init, double* red) {
    double v = init;
    for (int i = threadIdx.x; i < n; i += 256) v = red_op(v, part[i], op);
    return block_reduce(v, op, red);
}

// which: 0 = after INIT (pinned rows), 1 = top of an iteration (first: gap / pres come from the BUILD sweep), 2 = after AFF,
// 3 = after STEP, 4 = after UPBUILD
__global__ __launch_bounds__(256) void jq_ctrl(JArgs A, int which, int first) {
    const DevSession& S = A.S;
    const int mission = blockIdx.x, tid = threadIdx.x;
    const Ws w = carve(A, mission);
    __shared__ double red[8];
    double* st = w.st;
    if (st[ST_STATE] != 0.0 || (which >= 1 && st[ST_GO] != 0.0)) return;  // (which >= 1: a mission waiting for its polish is frozen)
    const JDims d = jdims(S.N, S.Mk[mission]);
    const int nsw = S.N * d.nch, npost = (d.nj * 3 * S.N + 255) / 256;
    const double nrows = st[ST_NROWS];
    if (which == 0) {
        const double pv = red_final(red_slot(w, A.L, RS_INIT, 0), nsw, 1, 0.0, red);
        if (tid == 0 && pv > 1e-6) st[ST_STATE] = 2.0, st[ST_REASON] = 1.0, st[ST_KKT] = pv;  // a constant (pinned) row is violated
    } else if (which == 1) {
        if (st[ST_RETRY] != 0.0) return;
        double gap = st[ST_GAP], pres = st[ST_PRES];
        if (first) {
            gap = red_final(red_slot(w, A.L, RS_BUILD, 0), nsw, 0, 0.0, red);
            pres = red_final(red_slot(w, A.L, RS_BUILD, 1), nsw, 1, 0.0, red);
        }
        const double dmax = red_final(red_slot(w, A.L, RS_POST, 0), npost, 1, 0.0, red);
        const double gmax = red_final(red_slot(w, A.L, RS_POST, 1), npost, 1, 0.0, red);
        if (tid == 0) {
            const double dres = dmax / (1.0 + gmax), mu = gap / nrows;
            st[ST_GAP] = gap, st[ST_PRES] = pres, st[ST_DRES] = dres, st[ST_MU] = mu;
            // dual (proximal) regularisation: the weights of this iteration's Newton matrix were built with DREGN by the last sweep
            st[ST_NREF] = mu < 1e-8 ? 2.0 : (mu < 1e-5 ? 1.0 : 0.0);
            st[ST_DREG] = st[ST_DREGN];
            st[ST_DREGN] = A.dreg_mode == 0 ? JQ_DREG : fmin(A.dreg_max, fmax(JQ_DREG, A.dreg_scale * mu));
            st[ST_KKT] = fmax(pres, fmax(dres, mu));
            // the exits of qp.hip, the third one at mu < 1e-9 instead of 1e-14: the explicit inverses of this solver put a floor under the
            // dual residual that RISES with the Newton weights (1e-9 .. 1e-5 once mu < 1e-10, even with two refinement steps per solve),
            // so going on only loses accuracy; what turns this iterate into the optimum is the active-set polish, not more iterations
            // CONTINUATION (round 5).  A final polish attempt that is refused does not end the mission any more: the iterate is untouched, so
            // the interior-point method goes on for another decade of mu (exit threshold exit_mu * 10^-ST_CONT) and the polish is tried
            // again on the sharper iterate -- the candidate set of a mu = 1e-10 iterate holds fewer rows that merely look active --, up to
            // JQ_MAX_CONT times; the safeguard below still takes back a step that loses the dual residual, and that ends the mission.
            const int cont = (int)st[ST_CONT];
            const double exit_mu = A.exit_mu * (cont == 0 ? 1.0 : (cont == 1 ? 1e-1 : (cont == 2 ? 1e-2 : 1e-3)));
            const bool ok = cont == 0 ? (pres < 1e-9 && dres < 1e-9 && mu < 1e-10) || (pres < 1e-6 && dres < 1e-9 && mu < 1e-13) ||
                                            (pres < 1e-9 && dres < 1e-7 && mu < exit_mu)
                                      : (pres < 1e-9 && dres < 1e-7 && mu < exit_mu);
            const bool polish_on = S.p.polish != 0;
            // SAFEGUARD.  Past mu ~ 1e-8 a step can cost the dual residual five orders of magnitude (explicit inverses at Newton weights
            // of 1e9: 5e-9 -> 1e-5 -> 1e-2 in two iterations, after which the method crawls for a hundred iterations or never returns).
            // If the iterate before the last step was acceptable and this one is not an exit and has lost the dual residual, the step is
            // taken back (x -= alpha dx by jq_unstep, the old (s, z) are still in the other parity) and that iterate is the answer
            // (final polish attempt, else unpolished with its own residuals).
            const bool lost = !ok && st[ST_PACC] != 0.0 && (dres >= 1e-7 || dres > 100.0 * st[ST_PDRES]);
            if (lost) {
                st[ST_CONT] = JQ_MAX_CONT;  // (no continuation from a step that was taken back)
                st[ST_REVERT] = 1.0, st[ST_PAR] = 1.0 - st[ST_PAR];
                st[ST_MU] = st[ST_PMU], st[ST_PRES] = st[ST_PPRES], st[ST_DRES] = st[ST_PDRES];
                st[ST_KKT] = fmax(st[ST_PPRES], fmax(st[ST_PDRES], st[ST_PMU]));
                st[ST_PACC] = 0.0;
            } else {
                st[ST_PACC] = (pres < 1e-9 && dres < 1e-7 && mu < 5e-8) ? 1.0 : 0.0;
                st[ST_PMU] = mu, st[ST_PPRES] = pres, st[ST_PDRES] = dres;
            }
            if (ok || lost) {
                if (polish_on)
                    st[ST_GO] = 1.0, st[ST_FINAL] = 1.0;  // final crossover; the mission stays "running" until it has been tried
                else
                    st[ST_STATE] = 1.0;
            } else if (!(gap == gap) || st[ST_ITER] >= JQ_MAX_ITERS) {
                st[ST_STATE] = 2.0, st[ST_REASON] = st[ST_BADPIV] != 0.0 ? 2.0 : 3.0;  // iteration cap or NaN (2: after a non-positive pivot of a knot's Schur complement)
            } else {
                // early crossover (qp.hip): once the active set shows (A.early_mu: see launch_planner_joint)
                const int tries = (int)st[ST_TRIES];
                if (polish_on && tries < 2 && pres < 1e-6 && dres < 1e-6 && mu < A.early_mu[tries]) st[ST_GO] = 1.0, st[ST_TRIES] = tries + 1.0;
                st[ST_ITER] += 1.0;
            }
        }
    } else if (which == 2) {
        if (st[ST_RETRY] != 0.0) return;
        const double vm = red_final(red_slot(w, A.L, RS_AFF, 0), nsw, 1, 1.0, red);
        const double q0 = red_final(red_slot(w, A.L, RS_AFF, 1), nsw, 0, 0.0, red);
        const double q1 = red_final(red_slot(w, A.L, RS_AFF, 2), nsw, 0, 0.0, red);
        const double q2 = red_final(red_slot(w, A.L, RS_AFF, 3), nsw, 0, 0.0, red);
        if (tid == 0) {
            const double a_aff = 1.0 / vm, mu = st[ST_MU];
            const double mu_aff = (q0 + a_aff * q1 + a_aff * a_aff * q2) / nrows;
            double sigma = pow(mu_aff / mu, A.tune[2]);
            st[ST_SIGMU] = sigma * mu, st[ST_AAFF] = a_aff;
        }
    } else if (which == 3) {
        if (st[ST_RETRY] != 0.0) return;
        const double vm = red_final(red_slot(w, A.L, RS_STEP, 0), nsw, 1, A.tune[4], red);
        if (tid == 0) {
            const double ap = A.tune[4] / vm;
            st[ST_ALPHA] = ap, st[ST_APPLIED] = 0.0, st[ST_BT] = 0.0;
            // centrality corrector (first == 1: enabled): tried whenever the Mehrotra direction stops short
            st[ST_GOK] = 0.0, st[ST_AP] = ap, st[ST_ATR] = fmin(1.0, ap + A.gond[0]), st[ST_GACT] = (first && ap < A.gond[2]) ? 1.0 : 0.0;
        }
    } else if (which == 5) {  // after STEPG: keep the corrected direction if the step grows by a tenth of what was asked for
        if (st[ST_RETRY] != 0.0 || st[ST_GACT] == 0.0) return;
        const double vm = red_final(red_slot(w, A.L, RS_STEP, 0), nsw, 1, A.tune[4], red);
        if (tid == 0) {
            const double an = A.tune[4] / vm;
            if (an >= st[ST_AP] + A.gond[1] * A.gond[0]) st[ST_GOK] = 1.0, st[ST_ALPHA] = an;
            st[ST_GACT] = 0.0;
        }
    } else if (which == 4) {
        if (A.retry_only && st[ST_RETRY] == 0.0) return;
        const double gap = red_final(red_slot(w, A.L, RS_BUILD, 0), nsw, 0, 0.0, red);
        const double pres = red_final(red_slot(w, A.L, RS_BUILD, 1), nsw, 1, 0.0, red);
        const double pmin = red_final(red_slot(w, A.L, RS_UP, 0), nsw, 2, 1e300, red);
        if (tid == 0) {
            if (pmin >= A.tune[3] * gap / nrows || st[ST_BT] >= 40.0 || !(gap == gap)) {
                st[ST_PAR] = 1.0 - st[ST_PAR], st[ST_RETRY] = 0.0, st[ST_GAP] = gap, st[ST_PRES] = pres;
            } else {  // wide-neighbourhood test failed: repeat the update sweep from the (untouched) old state with 0.8 alpha
                st[ST_APPLIED] = st[ST_ALPHA], st[ST_ALPHA] *= 0.8, st[ST_RETRY] = 1.0, st[ST_BT] += 1.0;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// factorisation
// ------------------------------------------------------------------------------------------------------------------------
// chain geometry of a mission: mid = nj / 2; chain 0 eliminates knots 0 .. mid-1 upwards, chain 1 knots nj-1 .. mid+1 downwards
struct Chain {
    bool active;
    int jj;    // 0-based knot of this step
    int prev;  // knot whose inverse feeds the Schur update (-1: none)
};
__device__ __forceinline__ Chain chain_step(const JDims& d, int chain, int s, bool mid) {
    Chain c;
    const int m = d.nj / 2, nl = m, nr = d.nj - 1 - m;
    if (mid) {
        c.active = chain == 0, c.jj = m, c.prev = -1;
    } else if (chain == 0) {
        c.active = s < nl, c.jj = s, c.prev = s > 0 ? s - 1 : -1;
    } else {
        c.active = s < nr, c.jj = d.nj - 1 - s, c.prev = s > 0 ? d.nj - s : -1;
    }
    return c;
}
// ping-pong buffers of the sweep: X[0] = the knot's slot in `inv`, X[1] = the chain's scratch; pass t reads X[(t + p0) & 1] and writes
// the other; p0 = (number of passes) & 1 makes the last pass land in X[0]
__device__ __forceinline__ int sweep_parity0(const JArgs& A, int nblk) { return (A.sweep2 ? (nblk + 1) / 2 : nblk) & 1; }
__device__ __forceinline__ double* sweep_buf(const Ws& w, const JDims& d, const JLayout& L, int chain, int jj, int which) {
    return which == 0 ? w.inv + (size_t)jj * L.nkpS * L.nkpS : w.scr + (size_t)chain * L.nkpS * L.nkpS;
}

// what a sweep kernel works on: kind 0 = the Schur complement of a knot (step s of a chain / the middle knot), kind 1 = the polish's
// S_AA (jqp_polish.inc; order and buffers from the mission's polish record)
struct SweepCtx {
    bool active;
    int nblk;
    const double* src;  // X[(k + p0) & 1]
    double* dst;        // X[(k + p0 + 1) & 1]
    double* Pk;         // pivot inverse of step k
    double* Pn;         // ... of step k + 1 (look-ahead)
    double* P2;         // double step (k, k + 1): the three tiles P00, P10, P11 of the pivot block's inverse
    double* Y;          // panel (double step: two tiles per block row)
    double* bad;        // counter of non-positive pivots
    const double* G;    // polish (kind 1): the matrix before the sweep (tile-major): its diagonal scales the deletion threshold; else nullptr
};
__device__ __forceinline__ SweepCtx sweep_ctx(const JArgs& A, const Ws& w, const JDims& d, int kind, int s, int mid, int k, int chain);

// T_jj + Schur updates, written into the first sweep buffer (tiles I >= J).  One thread per 3x3 block (Ai, Bi) = ((a, k), (b, l)).
__global__ __launch_bounds__(256) void jq_prep(JArgs A, int s, int mid) {
    const DevSession& S = A.S;
    const int mission = blockIdx.z, chain = blockIdx.y + A.chain0;
    const Ws w = carve(A, mission);
    if (w.st[ST_STATE] != 0.0 || w.st[ST_RETRY] != 0.0 || w.st[ST_GO] != 0.0) return;
    const int N = S.N, M = S.Mk[mission], MS = S.M;
    const JDims d = jdims(N, M);
    const Chain c = chain_step(d, chain, s, mid != 0);
    if (!c.active) return;
    const int jj = c.jj, j = jj + 1, nblk = d.nblk, oq = d.oq, n3 = 3 * N;
    double* X = sweep_buf(w, d, A.L, chain, jj, sweep_parity0(A, nblk));
    const float* normals = S.rsfc_normal + (size_t)mission * S.npair * MS * 3;
    const int t = blockIdx.x * 256 + threadIdx.x;
    // identity padding (rows / columns nk .. nkp-1)
    const int npad = d.nkp - d.nk;
    for (int i = t; i < npad * d.nkp; i += gridDim.x * 256) {
        const int r = d.nk + i / d.nkp, cc = i % d.nkp;
        X[telem(nblk, r, cc)] = r == cc ? 1.0 : 0.0;
        if ((cc >> 6) == nblk - 1) X[telem(nblk, cc, r)] = r == cc ? 1.0 : 0.0;
    }
    if (t >= n3 * n3) return;
    const int Ai = t / n3, Bi = t % n3;
    if ((3 * Ai + 2) / JT < (3 * Bi) / JT) return;  // no entry of this block lies in a tile I >= J
    const int a = Ai / 3, k = Ai % 3, b = Bi / 3, l = Bi % 3;
    double Sv[6];
    if (a == b) {
        const int kk = k < l ? k : l, ll = k < l ? l : k, sym = kk == 0 ? ll : (kk == 1 ? 2 + ll : 5);
#pragma unroll
        for (int pp = 0; pp < 6; ++pp) Sv[pp] = acc_sum(w, d, sym, (size_t)a * oq + 6 * (j - 1) + 3 + pp);
    } else {
        const int lo = a < b ? a : b, hi = a < b ? b : a;
        const size_t pi = pair_index(N, lo, hi);
        const float* nl = normals + (pi * M + (j - 1)) * 3;
        const float* nr = normals + (pi * M + j) * 3;
        const double nnl = (double)nl[k] * (double)nl[l], nnr = (double)nr[k] * (double)nr[l];
#pragma unroll
        for (int pp = 0; pp < 6; ++pp) Sv[pp] = -w.pwgt[pi * oq + 6 * (j - 1) + 3 + pp] * (pp < 3 ? nnl : nnr);
    }
    const double* L = w.Lk + 9 * j;
    double out[9];
#pragma unroll
    for (int e = 0; e < 3; ++e)
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            double acc = Sv[0] * L[e] * L[f] + Sv[1] * L[3 + e] * L[3 + f] + Sv[2] * L[6 + e] * L[6 + f];
            if (e == f) acc += Sv[3 + e];
            if (Ai == Bi) acc += w.Dk[9 * j + 3 * e + f];
            out[3 * e + f] = acc;
        }
    // Schur updates: - Cm Inv_prev Cm'  with the 3x3 coupling block Cm (the same for every (agent, dim))
    for (int side = 0; side < 2; ++side) {
        int prev;
        if (mid)
            prev = side == 0 ? (jj > 0 ? jj - 1 : -1) : (jj + 1 < d.nj ? jj + 1 : -1);
        else
            prev = side == 0 ? c.prev : -1;
        if (prev < 0) continue;
        double Cm[9];  // Cm[e][e']: row component e of knot jj, column component e' of knot prev
        if (prev < jj) {
#pragma unroll
            for (int e = 0; e < 3; ++e)
#pragma unroll
                for (int e2 = 0; e2 < 3; ++e2) Cm[3 * e + e2] = w.Ek[9 * jj + 3 * e2 + e];  // C_{jj-1}[r][c] = Ek[9 jj + 3 c + r]
        } else {
#pragma unroll
            for (int e = 0; e < 3; ++e)
#pragma unroll
                for (int e2 = 0; e2 < 3; ++e2) Cm[3 * e + e2] = w.Ek[9 * (jj + 1) + 3 * e + e2];  // C_jj'[e][e'] = C_jj[e'][e]
        }
        const double* Ip = w.inv + (size_t)prev * A.L.nkpS * A.L.nkpS;
        double V[9];
#pragma unroll
        for (int e = 0; e < 3; ++e)
#pragma unroll
            for (int f = 0; f < 3; ++f) V[3 * e + f] = Ip[telem(nblk, 3 * Ai + e, 3 * Bi + f)];
        double CV[9];
#pragma unroll
        for (int e = 0; e < 3; ++e)
#pragma unroll
            for (int f = 0; f < 3; ++f) CV[3 * e + f] = Cm[3 * e] * V[f] + Cm[3 * e + 1] * V[3 + f] + Cm[3 * e + 2] * V[6 + f];
#pragma unroll
        for (int e = 0; e < 3; ++e)
#pragma unroll
            for (int f = 0; f < 3; ++f) out[3 * e + f] -= CV[3 * e] * Cm[3 * f] + CV[3 * e + 1] * Cm[3 * f + 1] + CV[3 * e + 2] * Cm[3 * f + 2];
    }
#pragma unroll
    for (int e = 0; e < 3; ++e)
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            const int r = 3 * Ai + e, cc = 3 * Bi + f;
            if ((r >> 6) >= (cc >> 6)) X[telem(nblk, r, cc)] = out[3 * e + f];
        }
}

// ---- 64 x 64 SPD inverse in LDS (256 threads): the same blocked sweep one level down, 16 x 16 sub-tiles on the MFMA, the diagonal
// sub-tile by Gauss-Jordan with the rows in lanes (v_readlane broadcasts).  In: Am = the SPD tile, leading dimension LDA.
// Out: Am = -(tile)^-1.  *bad is set when a pivot is not positive.
constexpr int LDA = 66;  // (ds_read_b64 of lane (i, g) at row i, column 4 kk + g: conflict free with 66)
__device__ __forceinline__ double rl(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
constexpr size_t JQ_UPDATE_LDS_EXTRA = 64 * sizeof(double) + 16;  // thr[JT] + the bad-pivot flag behind Am and the scratch
struct InvScratch {
    double Pi[4][16 * 18];  // per wave: its copy of the diagonal sub-tile, inverted in place
    double Yb[4][16 * 18];
    // (Z of wave w lives in Pi[w]: a wave is done with its copy of the pivot sub-tile when it stores Z, only wave kk's copy is needed
    // afterwards and wave kk stores no Z -- 9 KB less, which lets a third workgroup of jq_update onto a CU)
};
#define JQ_WSYNC()                                             \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
    } while (0)
// 16 x 16 Gauss-Jordan inverse in LDS by ONE wave, in place (leading dimension 18): lane (r = l & 15, g = l >> 4) owns the entries
// D[r][4g .. 4g+3] in registers and publishes them after every column step; what a step needs from other lanes -- the pivot, the
// pivot row's segment, the row's entry in the pivot column -- are same-address LDS reads (broadcasts: ~7 cycles per 8 bytes, no
// v_readlane chains).  LDS serves a wave's operations in order, so a step's reads see the previous step's writes.
// thr (or nullptr): per column, the pivot size at or below which the row is DELETED from the solve instead of eliminated: row and column
// become zero, i.e. the result is the inverse of the matrix without that row, with a zero row and column in its place.  The polish uses it
// for active rows that are linear combinations of the rows before them (five control points around a knot are functions of three
// variables: a trajectory that runs along a box face makes four or five bound rows of one agent and axis active at once); everything
// downstream of a zero column of the pivot inverse -- panel, update, the other sub-tiles -- stays zero by the sweep's own algebra.
__device__ __forceinline__ bool gj16_lds(double* D, int lane, const double* thr = nullptr) {
    const int r = lane & 15, g = lane >> 4;
    double v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = D[r * 18 + 4 * g + q];
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const double p = D[c * 18 + c], f = D[r * 18 + c];
        double pr[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) pr[q] = D[c * 18 + 4 * g + q];
        if (thr && p <= thr[c]) {  // (wave-uniform)
            if (r == c) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = 0.0;
            }
            if (g == c / 4) v[c % 4] = 0.0;
        } else {
            ok = ok && (p > 0.0);
            const double ip = fast_rcp(p), fi = f * ip;
            if (r == c) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = pr[q] * ip;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] -= fi * pr[q];
            }
            if (g == c / 4) v[c % 4] = r == c ? ip : -fi;
        }
        JQ_WSYNC();
#pragma unroll
        for (int q = 0; q < 4; ++q) D[r * 18 + 4 * g + q] = v[q];
        JQ_WSYNC();
    }
    return ok;
}
__device__ __forceinline__ void inv64_lds_inl(double* Am, InvScratch* sc, int* bad, const double* thr = nullptr) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lg = lane >> 4;
    for (int kk = 0; kk < 4; ++kk) {
        // every wave inverts its own copy of the diagonal sub-tile (no barrier between the inversion and the wave's panel product)
        double* Pi = sc->Pi[wave];
#pragma unroll
        for (int q = 0; q < 4; ++q) Pi[li * 18 + 4 * lg + q] = Am[(16 * kk + li) * LDA + 16 * kk + 4 * lg + q];
        JQ_WSYNC();
        const bool okp = gj16_lds(Pi, lane, thr ? thr + 16 * kk : nullptr);
        if (!okp && tid == 0) *bad = 1;
        // panel: row block i = wave: Z = A[i][kk] (old), Y = Z Pi'
        if (wave != kk) {
            d4 acc = d4{0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double av = Am[(16 * wave + li) * LDA + 16 * kk + 4 * q + lg];
                const double bv = Pi[li * 18 + 4 * q + lg];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sc->Yb[wave][(lg + 4 * r) * 18 + li] = acc[r];
                sc->Pi[wave][(lg + 4 * r) * 18 + li] = Am[(16 * wave + lg + 4 * r) * LDA + 16 * kk + li];
            }
        }
        __syncthreads();
        // update: row block i = wave, all four column blocks
        for (int jb = 0; jb < 4; ++jb) {
            double* At = Am + (16 * wave) * LDA + 16 * jb;
            if (wave == kk && jb == kk) {
#pragma unroll
                for (int r = 0; r < 4; ++r) At[(lg + 4 * r) * LDA + li] = -Pi[(lg + 4 * r) * 18 + li];
            } else if (jb == kk) {
#pragma unroll
                for (int r = 0; r < 4; ++r) At[(lg + 4 * r) * LDA + li] = sc->Yb[wave][(lg + 4 * r) * 18 + li];
            } else if (wave == kk) {
#pragma unroll
                for (int r = 0; r < 4; ++r) At[(lg + 4 * r) * LDA + li] = sc->Yb[jb][li * 18 + lg + 4 * r];
            } else {
                d4 acc = d4{0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double av = sc->Yb[wave][li * 18 + 4 * q + lg];
                    const double bv = sc->Pi[jb][li * 18 + 4 * q + lg];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) At[(lg + 4 * r) * LDA + li] -= acc[r];
            }
        }
        __syncthreads();
    }
}

// (out of line for the pivot kernels, which have a CU to themselves; jq_update inlines it so that ITS register budget applies)
__device__ void inv64_lds(double* Am, InvScratch* sc, int* bad, const double* thr = nullptr) { inv64_lds_inl(Am, sc, bad, thr); }

// Pbuf[chain][parity] <- symmetrised inverse of the SPD tile held (as its NEGATED inverse after inv64_lds) in Am
__device__ __forceinline__ void store_pivot_inverse(const double* Am, double* Pg) {
    for (int i = threadIdx.x; i < JTT; i += 256) {
        const int r = i >> 6, c = i & 63;
        Pg[i] = -0.5 * (Am[r * LDA + c] + Am[c * LDA + r]);
    }
}

// pivot tile of step k.  k = 0: first pivot of a knot; k > 0: only in the bulk schedule (many workgroups per launch), where the update
// kernel has no look-ahead (jq_update_bulk)
__global__ __launch_bounds__(256) void jq_pivot0(JArgs A, int kind, int s, int mid, int k) {
    const DevSession& S = A.S;
    const int mission = blockIdx.z, chain = blockIdx.y + A.chain0;
    const Ws w = carve(A, mission);
    const JDims d = jdims(S.N, S.Mk[mission]);
    const SweepCtx c = sweep_ctx(A, w, d, kind, s, mid, k, chain);
    if (!c.active || k >= c.nblk) return;
    __shared__ double Am[JT * LDA];
    __shared__ InvScratch sc;
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    const double* tkk = c.src + ((size_t)k * c.nblk + k) * JTT;  // pivot tile (k, k) as the steps before k left it
    for (int i = threadIdx.x; i < JTT; i += 256) Am[(i >> 6) * LDA + (i & 63)] = tkk[i];
    __shared__ double thr[JT];
    if (c.G && threadIdx.x < JT) thr[threadIdx.x] = A.pol_tau * w.st[ST_PTAU] * c.G[((size_t)k * c.nblk + k) * JTT + (size_t)threadIdx.x * (JT + 1)];
    __syncthreads();
    inv64_lds(Am, &sc, &bad, c.G ? thr : nullptr);
    store_pivot_inverse(Am, c.Pk);
    // a non-positive pivot (the matrix is SPD in exact arithmetic) is counted, not fatal: the sweep needs no square roots, and with
    // Newton weights of 1e9 the last interior-point iterations work at the edge of double precision
    if (bad && threadIdx.x == 0) *c.bad = 1.0;  // (a flag: the two chains' workgroups may both set it, never a read-modify-write)
}

// operand fragments of a 16-row block for v_mfma_f64_16x16x4_f64 over K = 64: lane (i, g) holds rows[i][16 ch + 4 g + q], ch, q = 0..3
// (the four MFMA steps of a 16-chunk use k = 4 g + q on both operands: a permutation of the summation index, see tile_nt in qp.hip).
// TR: the operand is the TRANSPOSE of the stored tile (rows of the operand are columns of the tile).
__device__ __forceinline__ void load_frag(const double* tile, int row0, bool tr, int li, int lg, d4 (&f)[4]) {
    if (!tr) {
        const double* p = tile + (size_t)(row0 + li) * JT + 4 * lg;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) f[ch] = *reinterpret_cast<const d4*>(p + 16 * ch);
    } else {
#pragma unroll
        for (int ch = 0; ch < 4; ++ch)
#pragma unroll
            for (int q = 0; q < 4; ++q) f[ch][q] = tile[(size_t)(16 * ch + 4 * lg + q) * JT + row0 + li];
    }
}

// panel of step k: Y_J = B_Jk P for every J != k  (B_Jk = tile (J, k) below the pivot, tile (k, J)' left of it)
__global__ __launch_bounds__(256) void jq_panel(JArgs A, int kind, int s, int mid, int k) {
    const DevSession& S = A.S;
    const int mission = blockIdx.z, chain = blockIdx.y + A.chain0, J = blockIdx.x;
    const Ws w = carve(A, mission);
    const JDims d = jdims(S.N, S.Mk[mission]);
    const SweepCtx c = sweep_ctx(A, w, d, kind, s, mid, k, chain);
    if (!c.active || J >= c.nblk || J == k) return;
    const int nblk = c.nblk;
    const double* X = c.src;
    const double* P = c.Pk;
    const bool tr = J < k;
    const double* Z = X + (tr ? (size_t)k * nblk + J : (size_t)J * nblk + k) * JTT;
    double* Y = c.Y + (size_t)J * JTT;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    d4 zf[4];
    load_frag(Z, 16 * wave, tr, li, lg, zf);  // rows 16 wave .. of Z_J
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) {
        d4 pf[4];
        load_frag(P, 16 * tj, false, li, lg, pf);  // P symmetric: Z P = Z P'
        d4 acc = d4{0, 0, 0, 0};
#pragma unroll
        for (int ch = 0; ch < 4; ++ch)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(zf[ch][q], pf[ch][q], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) Y[(size_t)(16 * wave + lg + 4 * r) * JT + 16 * tj + li] = acc[r];
    }
}


// Y_T = B_Tk P of step k (what jq_panel computes for row block T, the same instructions in the same order), written to LDS instead of
// memory: when JArgs::fuse_panel is set every workgroup of jq_update forms the panel rows it needs itself and the panel launch -- a dependent
// kernel boundary per 64 columns -- disappears from the look-ahead schedule (one extra 64^3 product per tile, P and B_Tk come from the L2).
// Decided per launch (JArgs::fuse_panel): it pays while a launch is a round or so of workgroups (64 agents 0.267 -> 0.245 s; one chain of a
// 256-agent mission, 666 tiles, 3.84 -> 3.75 s) and costs 5 % when both chains of that mission share the launches (1332 tiles).
__device__ __forceinline__ void panel_rows_to_lds(const double* X, const double* P, int nblk, int k, int T, double* Am, int wave, int li, int lg) {
    const bool tr = T < k;
    const double* Z = X + (tr ? (size_t)k * nblk + T : (size_t)T * nblk + k) * JTT;
    d4 zf[4];
    load_frag(Z, 16 * wave, tr, li, lg, zf);
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) {
        d4 pf[4];
        load_frag(P, 16 * tj, false, li, lg, pf);
        d4 acc = d4{0, 0, 0, 0};
#pragma unroll
        for (int ch = 0; ch < 4; ++ch)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(zf[ch][q], pf[ch][q], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) Am[(16 * wave + lg + 4 * r) * LDA + 16 * tj + li] = acc[r];
    }
}

// update of step k: every tile (I, J), I >= J, of the lower triangle
//   (k, k) <- -P        (I, k) <- Y_I        (k, J) <- Y_J'        else  B_IJ - Y_I B_Jk'
// The last step writes -(...) = the inverse itself, with both triangles.  Look-ahead: the workgroup of tile (k+1, k+1) inverts it.
__global__ __launch_bounds__(256, 3) void jq_update(JArgs A, int kind, int s, int mid, int k) {
    const DevSession& S = A.S;
    // Which tile this workgroup takes.  Workgroups start in the order x, then y: the chains of a launch are interleaved (so that both
    // chains' first tiles start in the first round), and the tile that carries the look-ahead inversion -- 20 us of dependent work
    // on top of its update, the longest job of the launch -- is taken by the FIRST workgroup of its chain instead of one in the middle of
    // the triangle (with hundreds of tiles per chain, a 256-agent mission, it used to start in the second or third round and the
    // launch ended with that inversion alone on the chip).  The arithmetic of a tile does not depend on who computes it.
    const int lin = (int)(blockIdx.y * gridDim.x + blockIdx.x), nchl = (int)gridDim.y;
    const int mission = blockIdx.z, chain = lin % nchl + A.chain0;
    int b = lin / nchl;
    const Ws w = carve(A, mission);
    const JDims d = jdims(S.N, S.Mk[mission]);
    const SweepCtx c = sweep_ctx(A, w, d, kind, s, mid, k, chain);
    const int nblk = c.nblk;
    if (!c.active || b >= nblk * (nblk + 1) / 2) return;
    if (k + 1 < nblk) {
        const int lookb = (k + 1) * (k + 2) / 2 + (k + 1);
        b = b == 0 ? lookb : (b == lookb ? 0 : b);
    }
    int I = (int)((sqrtf(8.0f * b + 1.0f) - 1.0f) * 0.5f);
    if (I * (I + 1) / 2 > b) I--;
    if ((I + 1) * (I + 2) / 2 <= b) I++;
    const int J = b - I * (I + 1) / 2;
    const double* X = c.src;
    double* Xn = c.dst;
    const double* P = c.Pk;
    const double* Yb = c.Y;
    const bool last = k == nblk - 1, look = !last && I == k + 1 && J == k + 1, fuse = A.fuse_panel != 0;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;  // this wave's 32 x 32 quadrant
    // LDS is DYNAMIC here (JQ_UPDATE_LDS bytes per launch): with a static size the compiler derives the occupancy from a 64 KB LDS and then
    // spends 248 registers; the CU has 160 KB, three workgroups of 52.7 KB fit, and the register budget has to follow (launch bounds)
    extern __shared__ double jq_update_lds[];
    double* Am = jq_update_lds;                                        // [JT * LDA]
    InvScratch& sc = *reinterpret_cast<InvScratch*>(Am + JT * LDA);
    double* thr = reinterpret_cast<double*>(&sc + 1);                  // [JT]
    int& bad = *reinterpret_cast<int*>(thr + JT);
    double* out = Xn + ((size_t)I * nblk + J) * JTT;
    double* outT = Xn + ((size_t)J * nblk + I) * JTT;
    const double sgn = last ? -1.0 : 1.0;
    if (I == k || J == k) {  // copies (through LDS for the transposed ones)
        const double* src = (I == k && J == k) ? P : (J == k ? Yb + (size_t)I * JTT : Yb + (size_t)J * JTT);
        const bool tr = (I == k && J != k);
        const double f = (I == k && J == k) ? -sgn : sgn;
        if (fuse && !(I == k && J == k))
            panel_rows_to_lds(X, P, nblk, k, J == k ? I : J, Am, wave, li, lg);
        else
            for (int i = tid; i < JTT; i += 256) Am[(i >> 6) * LDA + (i & 63)] = src[i];
        __syncthreads();
        for (int i = tid; i < JTT; i += 256) {
            const int r = i >> 6, cc = i & 63;
            const double v = f * (tr ? Am[cc * LDA + r] : Am[r * LDA + cc]);
            out[i] = v;
            if (last && I != J) outT[(size_t)cc * JT + r] = v;
        }
        return;
    }
    const bool trJ = J < k;
    const double* Zt = X + (trJ ? (size_t)k * nblk + J : (size_t)J * nblk + k) * JTT;
    const double* Yt = Yb + (size_t)I * JTT;
    const double* Ct = X + ((size_t)I * nblk + J) * JTT;
    double cv[2][2][4];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) cv[ti][tj][r] = Ct[(size_t)(32 * wr + 16 * ti + lg + 4 * r) * JT + 32 * wc + 16 * tj + li];
    if (fuse) {
        panel_rows_to_lds(X, P, nblk, k, I, Am, wave, li, lg);
        __syncthreads();
    }
    {
        // operands of ONE 16-column chunk at a time (as jq_update_bulk): the register budget of three workgroups per CU
        d4 acc[2][2];
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = d4{0, 0, 0, 0};
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            d4 yf[2], zf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (fuse) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) yf[t][q] = Am[(32 * wr + 16 * t + li) * LDA + 16 * ch + 4 * lg + q];
                } else {
                    yf[t] = *reinterpret_cast<const d4*>(Yt + (size_t)(32 * wr + 16 * t + li) * JT + 16 * ch + 4 * lg);
                }
                if (!trJ) {
                    zf[t] = *reinterpret_cast<const d4*>(Zt + (size_t)(32 * wc + 16 * t + li) * JT + 16 * ch + 4 * lg);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) zf[t][q] = Zt[(size_t)(16 * ch + 4 * lg + q) * JT + 32 * wc + 16 * t + li];
                }
            }
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(yf[ti][q], zf[tj][q], acc[ti][tj], 0, 0, 0);
        }
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) cv[ti][tj][r] = sgn * (cv[ti][tj][r] - acc[ti][tj][r]);
    }
    if (!(last && I != J) && !look) {
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(size_t)(32 * wr + 16 * ti + lg + 4 * r) * JT + 32 * wc + 16 * tj + li] = cv[ti][tj][r];
        return;
    }
    // through LDS: mirrored store of the last step / look-ahead inversion of the next pivot
    if (fuse) __syncthreads();  // (Y_I is still being read from Am by the other waves)
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) Am[(32 * wr + 16 * ti + lg + 4 * r) * LDA + 32 * wc + 16 * tj + li] = cv[ti][tj][r];
    if (tid == 0) bad = 0;
    __syncthreads();
    for (int i = tid; i < JTT; i += 256) {
        const int r = i >> 6, cc = i & 63;
        out[i] = Am[r * LDA + cc];
        if (last) outT[i] = Am[cc * LDA + r];
    }
    if (look) {
        if (c.G && tid < JT) thr[tid] = A.pol_tau * w.st[ST_PTAU] * c.G[((size_t)(k + 1) * nblk + (k + 1)) * JTT + (size_t)tid * (JT + 1)];
        __syncthreads();
        inv64_lds_inl(Am, &sc, &bad, c.G ? thr : nullptr);
        store_pivot_inverse(Am, c.Pn);
        if (bad && tid == 0) *c.bad = 1.0;
    }
}


// The same update for launches with thousands of tiles (many resident missions, or one 256-agent mission): no look-ahead -- the next pivot
// is inverted by jq_pivot0(k + 1) in a launch of its own --, hence no LDS (the look-ahead scratch costs every workgroup of jq_update 61 KB),
// operands of ONE 16-column chunk at a time and a register budget for three workgroups per CU.
__global__ __launch_bounds__(256, 3) void jq_update_bulk(JArgs A, int kind, int s, int mid, int k) {
    const DevSession& S = A.S;
    const int mission = blockIdx.z, chain = blockIdx.y + A.chain0;
    const Ws w = carve(A, mission);
    const JDims d = jdims(S.N, S.Mk[mission]);
    const SweepCtx c = sweep_ctx(A, w, d, kind, s, mid, k, chain);
    const int nblk = c.nblk;
    if (!c.active || (int)blockIdx.x >= nblk * (nblk + 1) / 2) return;
    int I = (int)((sqrtf(8.0f * blockIdx.x + 1.0f) - 1.0f) * 0.5f);
    if (I * (I + 1) / 2 > (int)blockIdx.x) I--;
    if ((I + 1) * (I + 2) / 2 <= (int)blockIdx.x) I++;
    const int J = blockIdx.x - I * (I + 1) / 2;
    const bool last = k == nblk - 1;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;
    double* out = c.dst + ((size_t)I * nblk + J) * JTT;
    double* outT = c.dst + ((size_t)J * nblk + I) * JTT;
    const double sgn = last ? -1.0 : 1.0;
    if (I == k || J == k) {
        const double* src = (I == k && J == k) ? c.Pk : (J == k ? c.Y + (size_t)I * JTT : c.Y + (size_t)J * JTT);
        const bool tr = (I == k && J != k);
        const double f = (I == k && J == k) ? -sgn : sgn;
        for (int i = tid; i < JTT; i += 256) {
            const int r = i >> 6, cc = i & 63;
            const double v = f * (tr ? src[(size_t)cc * JT + r] : src[i]);
            out[i] = v;
            if (last && I != J) outT[(size_t)cc * JT + r] = v;
        }
        return;
    }
    const bool trJ = J < k;
    const double* Zt = c.src + (trJ ? (size_t)k * nblk + J : (size_t)J * nblk + k) * JTT;
    const double* Yt = c.Y + (size_t)I * JTT;
    const double* Ct = c.src + ((size_t)I * nblk + J) * JTT;
    double cv[2][2][4];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) cv[ti][tj][r] = Ct[(size_t)(32 * wr + 16 * ti + lg + 4 * r) * JT + 32 * wc + 16 * tj + li];
    d4 acc[2][2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = d4{0, 0, 0, 0};
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
        d4 yf[2], zf[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            yf[t] = *reinterpret_cast<const d4*>(Yt + (size_t)(32 * wr + 16 * t + li) * JT + 16 * ch + 4 * lg);
            if (!trJ) {
                zf[t] = *reinterpret_cast<const d4*>(Zt + (size_t)(32 * wc + 16 * t + li) * JT + 16 * ch + 4 * lg);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) zf[t][q] = Zt[(size_t)(16 * ch + 4 * lg + q) * JT + 32 * wc + 16 * t + li];
            }
        }
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(yf[ti][q], zf[tj][q], acc[ti][tj], 0, 0, 0);
    }
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double v = sgn * (cv[ti][tj][r] - acc[ti][tj][r]);
                const int rr = 32 * wr + 16 * ti + lg + 4 * r, cc = 32 * wc + 16 * tj + li;
                out[(size_t)rr * JT + cc] = v;
                if (last && I != J) outT[(size_t)cc * JT + rr] = v;
            }
}

// ------------------------------------------------------------------------------------------------------------------------
// Double steps of the bulk schedule: the pivots (k, k + 1) in ONE pass over the matrix.  jq_update_bulk is bound by HBM, not by the MFMA
// (profiles/r05_joint_pmc.txt: 1.6 GB of reads and writes per 286 us launch = 5.6 TB/s at 200 resident missions): every pass reads and
// writes the knot's whole lower triangle for a rank-64 update.  Sweeping on the 128 x 128 pivot block [B_kk B_k+1,k'; B_k+1,k B_k+1,k+1]
// is the same two sweep steps composed -- P2 = block^-1, Y2_J = [B_Jk B_J,k+1] P2, (I, J) <- B_IJ - Y2_I [B_Jk B_J,k+1]' -- with half the
// passes over the matrix per unit of arithmetic.
// ------------------------------------------------------------------------------------------------------------------------
// P2 by block elimination in one workgroup: Pa = B_kk^-1, W = B_k+1,k Pa, Ps = (B_k+1,k+1 - W B_k+1,k')^-1,
// P11 = Ps, P10 = -Ps W, P00 = Pa + W' Ps W.  The four 64^3 products run on plain FMAs out of LDS (a few microseconds per knot and pass).
__global__ __launch_bounds__(256) void jq_pivot2(JArgs A, int s, int mid, int k) {
    const DevSession& S = A.S;
    const int mission = blockIdx.z, chain = blockIdx.y + A.chain0, tid = threadIdx.x;
    const Ws w = carve(A, mission);
    const JDims d = jdims(S.N, S.Mk[mission]);
    const SweepCtx c = sweep_ctx(A, w, d, 0, s, mid, k, chain);
    if (!c.active || k + 1 >= c.nblk) return;
    extern __shared__ double lds2[];  // Am, Bm, Wm: 3 x JT x LDA doubles
    __shared__ InvScratch sc;
    __shared__ int bad;
    double *Am = lds2, *Bm = lds2 + JT * LDA, *Wm = lds2 + 2 * JT * LDA;
    const int nblk = c.nblk;
    const double* t00 = c.src + ((size_t)k * nblk + k) * JTT;
    const double* t10 = c.src + ((size_t)(k + 1) * nblk + k) * JTT;
    const double* t11 = c.src + ((size_t)(k + 1) * nblk + k + 1) * JTT;
    double *P00 = c.P2, *P10 = c.P2 + JTT, *P11 = c.P2 + 2 * JTT;
    if (tid == 0) bad = 0;
    for (int i = tid; i < JTT; i += 256) Am[(i >> 6) * LDA + (i & 63)] = t00[i], Bm[(i >> 6) * LDA + (i & 63)] = t10[i];
    __syncthreads();
    inv64_lds(Am, &sc, &bad);  // Am = -Pa
    double pa[16];
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        const int i = tid + 256 * n, r = i >> 6, cc = i & 63;
        pa[n] = -0.5 * (Am[r * LDA + cc] + Am[cc * LDA + r]);
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        const int i = tid + 256 * n;
        Am[(i >> 6) * LDA + (i & 63)] = pa[n];
    }
    __syncthreads();
    const int r0 = 4 * (tid >> 4), c0 = 4 * (tid & 15);
    double acc[4][4];
    // MODE 0: X[r][m] Yv[m][c]   1: X[r][m] Yv[c][m]   2: X[m][r] Yv[m][c]
#define JQ_MM(MODE, X, Yv)                                                                            \
    do {                                                                                              \
        _Pragma("unroll") for (int x = 0; x < 4; ++x) _Pragma("unroll") for (int y = 0; y < 4; ++y) acc[x][y] = 0.0; \
        for (int m = 0; m < JT; ++m) {                                                                \
            double a[4], b[4];                                                                        \
            _Pragma("unroll") for (int x = 0; x < 4; ++x) a[x] = (MODE) == 2 ? (X)[m * LDA + r0 + x] : (X)[(r0 + x) * LDA + m]; \
            _Pragma("unroll") for (int y = 0; y < 4; ++y) b[y] = (MODE) == 1 ? (Yv)[(c0 + y) * LDA + m] : (Yv)[m * LDA + c0 + y]; \
            _Pragma("unroll") for (int x = 0; x < 4; ++x) _Pragma("unroll") for (int y = 0; y < 4; ++y) acc[x][y] = fma(a[x], b[y], acc[x][y]); \
        }                                                                                             \
    } while (0)
    JQ_MM(0, Bm, Am);  // W = B10 Pa
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) Wm[(r0 + x) * LDA + c0 + y] = acc[x][y];
    __syncthreads();
    JQ_MM(1, Wm, Bm);  // W B10'
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) Am[(r0 + x) * LDA + c0 + y] = t11[(size_t)(r0 + x) * JT + c0 + y] - acc[x][y];
    __syncthreads();
    inv64_lds(Am, &sc, &bad);  // Am = -Ps
    double ps[16];
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        const int i = tid + 256 * n, r = i >> 6, cc = i & 63;
        ps[n] = -0.5 * (Am[r * LDA + cc] + Am[cc * LDA + r]);
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        const int i = tid + 256 * n;
        Am[(i >> 6) * LDA + (i & 63)] = ps[n];
        P11[i] = ps[n];
    }
    __syncthreads();
    JQ_MM(0, Am, Wm);  // V = Ps W
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) Bm[(r0 + x) * LDA + c0 + y] = acc[x][y], P10[(size_t)(r0 + x) * JT + c0 + y] = -acc[x][y];
    __syncthreads();
    JQ_MM(2, Wm, Bm);  // W' V
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) Am[(r0 + x) * LDA + c0 + y] = acc[x][y];
    __syncthreads();
#undef JQ_MM
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        const int i = tid + 256 * n, r = i >> 6, cc = i & 63;
        P00[i] = pa[n] + 0.5 * (Am[r * LDA + cc] + Am[cc * LDA + r]);
    }
    if (bad && tid == 0) *c.bad = 1.0;
}

// panel of the double step: Y2_J = [B_Jk B_J,k+1] P2 for every block row J outside the pivot block (two tiles per J)
__global__ __launch_bounds__(256) void jq_panel2(JArgs A, int s, int mid, int k) {
    const DevSession& S = A.S;
    const int mission = blockIdx.z, chain = blockIdx.y + A.chain0, J = blockIdx.x;
    const Ws w = carve(A, mission);
    const JDims d = jdims(S.N, S.Mk[mission]);
    const SweepCtx c = sweep_ctx(A, w, d, 0, s, mid, k, chain);
    const int nblk = c.nblk;
    if (!c.active || J >= nblk || J == k || J == k + 1 || k + 1 >= nblk) return;
    const bool tr = J < k;
    const double* Z0 = c.src + (tr ? (size_t)k * nblk + J : (size_t)J * nblk + k) * JTT;
    const double* Z1 = c.src + (tr ? (size_t)(k + 1) * nblk + J : (size_t)J * nblk + k + 1) * JTT;
    const double *P00 = c.P2, *P10 = c.P2 + JTT, *P11 = c.P2 + 2 * JTT;
    double* Y0 = c.Y + (size_t)J * 2 * JTT;
    double* Y1 = Y0 + JTT;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    d4 z0[4], z1[4];
    load_frag(Z0, 16 * wave, tr, li, lg, z0);
    load_frag(Z1, 16 * wave, tr, li, lg, z1);
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) {
        d4 pf[4];
        d4 a0 = d4{0, 0, 0, 0}, a1 = d4{0, 0, 0, 0};
        load_frag(P00, 16 * tj, false, li, lg, pf);  // Z0 P00 (symmetric)
#pragma unroll
        for (int ch = 0; ch < 4; ++ch)
#pragma unroll
            for (int q = 0; q < 4; ++q) a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(z0[ch][q], pf[ch][q], a0, 0, 0, 0);
        load_frag(P10, 16 * tj, true, li, lg, pf);  // Z1 P10: the operand's rows are P10's columns
#pragma unroll
        for (int ch = 0; ch < 4; ++ch)
#pragma unroll
            for (int q = 0; q < 4; ++q) a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(z1[ch][q], pf[ch][q], a0, 0, 0, 0);
        load_frag(P10, 16 * tj, false, li, lg, pf);  // Z0 P01 = Z0 P10'
#pragma unroll
        for (int ch = 0; ch < 4; ++ch)
#pragma unroll
            for (int q = 0; q < 4; ++q) a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(z0[ch][q], pf[ch][q], a1, 0, 0, 0);
        load_frag(P11, 16 * tj, false, li, lg, pf);  // Z1 P11 (symmetric)
#pragma unroll
        for (int ch = 0; ch < 4; ++ch)
#pragma unroll
            for (int q = 0; q < 4; ++q) a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(z1[ch][q], pf[ch][q], a1, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            Y0[(size_t)(16 * wave + lg + 4 * r) * JT + 16 * tj + li] = a0[r];
            Y1[(size_t)(16 * wave + lg + 4 * r) * JT + 16 * tj + li] = a1[r];
        }
    }
}

// update of the double step: every tile (I, J), I >= J, of the lower triangle
//   pivot block <- -P2      (I, k + h) <- Y2_I[h]      (k + h, J) <- Y2_J[h]'      else  B_IJ - Y2_I[0] B_Jk' - Y2_I[1] B_J,k+1'
// (the last pass writes -(...) = the inverse itself, with both triangles)
__global__ __launch_bounds__(256, 3) void jq_update2_bulk(JArgs A, int s, int mid, int k) {
    const DevSession& S = A.S;
    const int mission = blockIdx.z, chain = blockIdx.y + A.chain0;
    const Ws w = carve(A, mission);
    const JDims d = jdims(S.N, S.Mk[mission]);
    const SweepCtx c = sweep_ctx(A, w, d, 0, s, mid, k, chain);
    const int nblk = c.nblk;
    if (!c.active || k + 1 >= nblk || (int)blockIdx.x >= nblk * (nblk + 1) / 2) return;
    int I = (int)((sqrtf(8.0f * blockIdx.x + 1.0f) - 1.0f) * 0.5f);
    if (I * (I + 1) / 2 > (int)blockIdx.x) I--;
    if ((I + 1) * (I + 2) / 2 <= (int)blockIdx.x) I++;
    const int J = blockIdx.x - I * (I + 1) / 2;
    const bool last = k + 1 == nblk - 1;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;
    double* out = c.dst + ((size_t)I * nblk + J) * JTT;
    double* outT = c.dst + ((size_t)J * nblk + I) * JTT;
    const double sgn = last ? -1.0 : 1.0;
    const bool Ik = I == k || I == k + 1, Jk = J == k || J == k + 1;
    if (Ik || Jk) {
        const double* src;
        bool tr = false;
        double f = sgn;
        if (Ik && Jk)
            src = c.P2 + (size_t)(I == k ? 0 : (J == k ? 1 : 2)) * JTT, f = -sgn;
        else if (Jk)
            src = c.Y + ((size_t)I * 2 + (J - k)) * JTT;
        else
            src = c.Y + ((size_t)J * 2 + (I - k)) * JTT, tr = true;
        for (int i = tid; i < JTT; i += 256) {
            const int r = i >> 6, cc = i & 63;
            const double v = f * (tr ? src[(size_t)cc * JT + r] : src[i]);
            out[i] = v;
            if (last && I != J) outT[(size_t)cc * JT + r] = v;
        }
        return;
    }
    const bool trJ = J < k;
    const double* Ct = c.src + ((size_t)I * nblk + J) * JTT;
    double cv[2][2][4];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) cv[ti][tj][r] = Ct[(size_t)(32 * wr + 16 * ti + lg + 4 * r) * JT + 32 * wc + 16 * tj + li];
    d4 acc[2][2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = d4{0, 0, 0, 0};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const double* Zt = c.src + (trJ ? (size_t)(k + h) * nblk + J : (size_t)J * nblk + k + h) * JTT;
        const double* Yt = c.Y + ((size_t)I * 2 + h) * JTT;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            d4 yf[2], zf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                yf[t] = *reinterpret_cast<const d4*>(Yt + (size_t)(32 * wr + 16 * t + li) * JT + 16 * ch + 4 * lg);
                if (!trJ) {
                    zf[t] = *reinterpret_cast<const d4*>(Zt + (size_t)(32 * wc + 16 * t + li) * JT + 16 * ch + 4 * lg);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) zf[t][q] = Zt[(size_t)(16 * ch + 4 * lg + q) * JT + 32 * wc + 16 * t + li];
                }
            }
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(yf[ti][q], zf[tj][q], acc[ti][tj], 0, 0, 0);
        }
    }
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double v = sgn * (cv[ti][tj][r] - acc[ti][tj][r]);
                const int rr = 32 * wr + 16 * ti + lg + 4 * r, cc = 32 * wc + 16 * tj + li;
                out[(size_t)rr * JT + cc] = v;
                if (last && I != J) outT[(size_t)cc * JT + rr] = v;
            }
}

// ------------------------------------------------------------------------------------------------------------------------
// substitutions with the stored inverses.  mode 0 forward (w_j = S_j^-1 (r_j - C w_prev)), 1 middle (x_m = S_m^-1 (r_m - C w_l -
// C' w_r)), 2 backward (x_j = w_j - S_j^-1 C' x_next).  The solution replaces rhs.  One workgroup per 16-row slab.
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void jq_mv(JArgs A, int mode, int s) {
    const DevSession& S = A.S;
    const int mission = blockIdx.z, chain = blockIdx.y + A.chain0, tid = threadIdx.x;
    const Ws w = carve(A, mission);
    if (w.st[ST_STATE] != 0.0 || w.st[ST_RETRY] != 0.0 || w.st[ST_GO] != 0.0) return;
    const JDims d = jdims(S.N, S.Mk[mission]);
    const Chain c = chain_step(d, chain, s, mode == 1 || mode == 3);
    if (!c.active || (int)blockIdx.x * 16 >= d.nkp) return;
    if (w.st[ST_NREF] < (double)A.ref_gate) return;  // a refinement pass this mission did not ask for
    if (A.gond_only && w.st[ST_GACT] == 0.0) return;
    if (mode == 3) {  // x_m: wv -> rhs (the middle step cannot write rhs itself: the other slabs of its launch still read r_m there)
        const int row = blockIdx.x * 16 + (tid >> 4);
        if ((tid & 15) == 0 && row < d.nk) w.rhs[(size_t)c.jj * d.nkp + row] = w.wv[(size_t)c.jj * d.nkp + row];
        return;
    }
    const int jj = c.jj, nkp = d.nkp, nk = d.nk, nblk = d.nblk;
    extern __shared__ double vsh[];  // nkp
    const double* rj = w.rhs + (size_t)jj * nkp;
    for (int i = tid; i < nkp; i += 256) {
        double v = 0;
        if (i < nk) {
            const int u3 = (i / 3) * 3, e = i % 3;
            if (mode == 0) {
                v = rj[i];
                if (c.prev >= 0) {
                    const double* wp = w.wv + (size_t)c.prev * nkp + u3;
                    if (c.prev < jj)
                        v -= w.Ek[9 * jj + e] * wp[0] + w.Ek[9 * jj + 3 + e] * wp[1] + w.Ek[9 * jj + 6 + e] * wp[2];
                    else
                        v -= w.Ek[9 * (jj + 1) + 3 * e] * wp[0] + w.Ek[9 * (jj + 1) + 3 * e + 1] * wp[1] + w.Ek[9 * (jj + 1) + 3 * e + 2] * wp[2];
                }
            } else if (mode == 1) {
                v = rj[i];
                if (jj > 0) {
                    const double* wp = w.wv + (size_t)(jj - 1) * nkp + u3;
                    v -= w.Ek[9 * jj + e] * wp[0] + w.Ek[9 * jj + 3 + e] * wp[1] + w.Ek[9 * jj + 6 + e] * wp[2];
                }
                if (jj + 1 < d.nj) {
                    const double* wp = w.wv + (size_t)(jj + 1) * nkp + u3;
                    v -= w.Ek[9 * (jj + 1) + 3 * e] * wp[0] + w.Ek[9 * (jj + 1) + 3 * e + 1] * wp[1] + w.Ek[9 * (jj + 1) + 3 * e + 2] * wp[2];
                }
            } else {
                // the knot towards the middle: chain 0 -> jj + 1, chain 1 -> jj - 1 (its solution is already in rhs)
                if (chain == 0) {
                    const double* xp = w.rhs + (size_t)(jj + 1) * nkp + u3;
                    v = w.Ek[9 * (jj + 1) + 3 * e] * xp[0] + w.Ek[9 * (jj + 1) + 3 * e + 1] * xp[1] + w.Ek[9 * (jj + 1) + 3 * e + 2] * xp[2];
                } else {
                    const double* xp = w.rhs + (size_t)(jj - 1) * nkp + u3;
                    v = w.Ek[9 * jj + e] * xp[0] + w.Ek[9 * jj + 3 + e] * xp[1] + w.Ek[9 * jj + 6 + e] * xp[2];
                }
            }
        }
        vsh[i] = v;
    }
    __syncthreads();
    const double* Inv = w.inv + (size_t)jj * A.L.nkpS * A.L.nkpS;
    const int row = blockIdx.x * 16 + (tid >> 4), l16 = tid & 15;
    const double* rp = Inv + ((size_t)(row >> 6) * nblk) * JTT + (size_t)(row & 63) * JT + 4 * l16;
    double acc = 0;
    for (int Jt = 0; Jt < nblk; ++Jt) {
        const d4 m = *reinterpret_cast<const d4*>(rp + (size_t)Jt * JTT);
        const double* vv = vsh + 64 * Jt + 4 * l16;
        acc += m[0] * vv[0] + m[1] * vv[1] + m[2] * vv[2] + m[3] * vv[3];
    }
    acc += __shfl_xor(acc, 8), acc += __shfl_xor(acc, 4), acc += __shfl_xor(acc, 2), acc += __shfl_xor(acc, 1);
    if (l16 == 0 && row < nk) {
        if (mode == 0)
            w.wv[(size_t)jj * nkp + row] = acc;
        else if (mode == 1)
            w.wv[(size_t)jj * nkp + row] = acc;  // (copied to rhs by the mode 3 launch)
        else
            w.rhs[(size_t)jj * nkp + row] = w.wv[(size_t)jj * nkp + row] - acc;
    }
}


// ------------------------------------------------------------------------------------------------------------------------
// two-rank factorisation (JointShard): what one rank's chain produces and the other rank needs, packed per mission into a contiguous
// buffer (dir 0, chain = this rank's) / scattered from the peer's buffer (dir 1, chain = the peer's).  what 0: the explicit inverse
// of the chain's last knot (nkp^2 doubles; jq_prep of the middle knot reads both neighbours'); 1: that knot's forward vector w (nkp;
// the middle solve, jq_mv mode 1, reads both); 2: the chain's rows of the solution in rhs (slot = (njS / 2) nkp).  Bytes only --
// nothing is added, so both ranks hold bit-identical vectors afterwards.  The gates are those of the kernels that produced the data.
// ------------------------------------------------------------------------------------------------------------------------
// Every exchange starts with a header of JQ_XHDR doubles written by the host (launch_planner_joint): the two ranks run the same host
// loop on replicated state, and nothing else keeps their send / recv pairs matched -- so each exchange carries its sequence number, its
// kind, its payload size and a hash of the state words the rank polled last, all compared with the rank's own after the hook returns;
// a mismatch (the ranks have diverged: the payload would be unpacked into the wrong slot) ends the run with RBP_ERR_EXCHANGE instead of
// a hang or silently wrong data.  Slot 5 is the poison word: a rank that fails locally sends it so that its peer stops too.
#define JQ_XHDR 8
#define JQ_XMAGIC 1380077656.0 /* "RBPX" */
// the same header for the stream-ordered exchange (JointShard::exchange_stream): written into the send buffer and compared with the peer's
// in the receive buffer BY KERNELS, in stream order -- no host synchronisation per exchange.  A mismatch is recorded in xerr (first one
// wins) and every later unpack of the run is skipped (jq_xfer reads xerr); the host finds it at the next per-round poll.
__global__ void jq_xhdr_write(double* send, double seq, double what, double bytes, double hash) {
    if (threadIdx.x == 0)
        send[0] = JQ_XMAGIC, send[1] = seq, send[2] = what, send[3] = bytes, send[4] = hash, send[5] = 0.0, send[6] = 0.0, send[7] = 0.0;
}
__global__ void jq_xhdr_check(const double* recv, double seq, double what, double bytes, double hash, double* xerr) {
    if (threadIdx.x != 0 || xerr[0] != 0.0) return;
    const double mine[6] = {JQ_XMAGIC, seq, what, bytes, hash, 0.0};
    const int order[6] = {0, 5, 1, 2, 3, 4};
    for (int oi = 0; oi < 6; ++oi)
        if (recv[order[oi]] != mine[order[oi]]) {
            xerr[0] = (double)(order[oi] + 1), xerr[1] = seq;
            return;
        }
}
__global__ __launch_bounds__(256) void jq_xfer(JArgs A, int what, int dir, int chain, double* buf, const double* xerr) {
    const DevSession& S = A.S;
    const int mission = blockIdx.y;
    if (dir == 1 && xerr && xerr[0] != 0.0) return;  // (stream-ordered exchange: the header did not match -- what lies in buf is not ours to unpack)
    const Ws w = carve(A, mission);
    if (w.st[ST_STATE] != 0.0 || w.st[ST_RETRY] != 0.0 || w.st[ST_GO] != 0.0) return;
    if (what != 0) {
        if (w.st[ST_NREF] < (double)A.ref_gate) return;
        if (A.gond_only && w.st[ST_GACT] == 0.0) return;
    }
    const JDims d = jdims(S.N, S.Mk[mission]);
    const int m = d.nj / 2, nsteps = chain == 0 ? m : d.nj - 1 - m;  // knots of this chain
    if (nsteps <= 0) return;
    const size_t nkp = d.nkp, nkp2 = nkp * nkp;
    const int last = chain == 0 ? m - 1 : m + 1, first = chain == 0 ? 0 : m + 1;
    double* dev;
    size_t n, slot;
    if (what == 0)
        dev = w.inv + (size_t)last * A.L.nkpS * A.L.nkpS, n = nkp2, slot = (size_t)A.L.nkpS * A.L.nkpS + 1;  // (+ 1: the chain's bad-pivot flag)
    else if (what == 1)
        dev = w.wv + (size_t)last * nkp, n = nkp, slot = A.L.nkpS;
    else
        dev = w.rhs + (size_t)first * nkp, n = (size_t)nsteps * nkp, slot = (size_t)(A.L.njS / 2) * A.L.nkpS;
    double* b = buf + (size_t)mission * slot;
    if (what == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
        // a non-positive pivot seen by ONE chain's rank must reach the other: ST_BADPIV feeds ST_REASON, which is replicated state
        // (both ranks report the same failure reason).  Set-only on the receiving side, like the sweeps' own plain stores.
        if (dir == 0)
            b[slot - 1] = w.st[ST_BADPIV];
        else if (b[slot - 1] != 0.0)
            w.st[ST_BADPIV] = 1.0;
    }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (dir == 0)
            b[i] = dev[i];
        else
            dev[i] = b[i];
    }
}

#define JQ_POLISH_PART 2
#include "jqp_polish.inc"
#undef JQ_POLISH_PART

__device__ __forceinline__ SweepCtx sweep_ctx(const JArgs& A, const Ws& w, const JDims& d, int kind, int s, int mid, int k, int chain) {
    SweepCtx c;
    c.bad = w.st + ST_BADPIV;
    c.G = nullptr;
    if (kind == 0) {
        const Chain ch = chain_step(d, chain, s, mid != 0);
        c.active = ch.active && w.st[ST_STATE] == 0.0 && w.st[ST_RETRY] == 0.0 && w.st[ST_GO] == 0.0;
        c.nblk = d.nblk;
        // pass t of np: a double step takes the pivots (2 t, 2 t + 1), an odd order ends with a single step
        const int t = A.sweep2 ? (k >> 1) : k, p0 = sweep_parity0(A, d.nblk);
        c.src = sweep_buf(w, d, A.L, chain, ch.jj, (t + p0) & 1);
        c.dst = sweep_buf(w, d, A.L, chain, ch.jj, (t + p0 + 1) & 1);
        c.Pk = w.P + ((size_t)chain * 2 + (k & 1)) * JTT, c.Pn = w.P + ((size_t)chain * 2 + ((k + 1) & 1)) * JTT;
        c.P2 = w.P + (size_t)(4 + 3 * chain) * JTT;
        c.Y = w.Y + (size_t)chain * 2 * A.L.nblkS * JTT;
    } else {
        const Pol p = pol_carve(A, blockIdx.z);
        c.nblk = p.cnt[PC_NBLK];
        c.active = chain == 0 && w.st[ST_STATE] == 0.0 && w.st[ST_GO] != 0.0 && w.st[ST_PSTATE] == (double)PS_SOLVE && !p.cnt[PC_BPPDONE] && k < c.nblk;
        const int p0 = c.nblk & 1;
        c.src = ((k + p0) & 1) ? p.W1 : p.W0;
        c.dst = ((k + p0 + 1) & 1) ? p.W1 : p.W0;
        c.Pk = p.P + (size_t)(k & 1) * JTT, c.Pn = p.P + (size_t)((k + 1) & 1) * JTT;
        c.Y = p.Y;
        c.P2 = nullptr;
        c.G = p.G;
    }
    return c;
}

// ------------------------------------------------------------------------------------------------------------------------
// end of a solve: objective sum x' Q_p x (cplex.getObjValue, rbp_planner.hpp:164), diagnostics, status
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void jq_finish(JArgs A) {
    const DevSession& S = A.S;
    const int mission = blockIdx.x, tid = threadIdx.x;
    const Ws w = carve(A, mission);
    __shared__ double red[8];
    const int N = S.N, M = S.Mk[mission], MS = S.M, oq = 6 * M;
    if (S.status[mission] != 0) return;
    double* scal = S.scalars + (size_t)mission * SC_N;
    if (w.st[ST_STATE] != 1.0) {
        if (tid == 0) {
            atomicCAS(&S.status[mission], 0, (int)RBP_ERR_QP_FAILED);
            scal[SC_PROF0 + 1] = w.st[ST_REASON], scal[SC_PROF0 + 2] = w.st[ST_ITER];
        }
        return;
    }
    const double* ctrl = S.ctrl + (size_t)mission * N * 3 * 6 * MS;
    double obj = 0;
    for (int it = tid; it < N * 3 * M; it += 256) {
        const int a = it / (3 * M), k = (it / M) % 3, m = it % M;
        const double* xs = ctrl + ((size_t)a * 3 + k) * oq + 6 * m;
        double q = 0;
        for (int i = 0; i < 6; ++i)
            for (int jj = 0; jj < 6; ++jj) q += jc_Qbase[6 * i + jj] * xs[i] * xs[jj];
        obj += q * w.segsc[m];
    }
    obj = block_reduce(obj, 0, red);
    if (tid == 0) {
        scal[SC_TOTAL_COST] = obj;
        scal[SC_IPM_ITERS] += w.st[ST_ITER];
        scal[SC_QP_SOLVED] += 1;
        scal[SC_POLISHED] += w.st[ST_POLISHED];
        if (w.st[ST_POLISHED] == 0.0) scal[SC_PROF0] = w.st[ST_REASON];  // why the polish was refused (100 + jqp_polish.inc reason)
        scal[SC_KKT_MAX] = fmax(scal[SC_KKT_MAX], w.st[ST_KKT]);
        scal[SC_FLOPS] += w.st[ST_FLOPS];
        scal[SC_ROWS] += w.st[ST_NROWS] * (1.0 + 4.0 * w.st[ST_ITER]);
    }
}

// logged flops of one factorisation (the MFMA work actually issued: panels and rank-64 updates of the lower triangle) and of the two
// solves of an iteration
__global__ void jq_count(JArgs A) {
    const int mission = blockIdx.x;
    const Ws w = carve(A, mission);
    if (threadIdx.x != 0 || w.st[ST_STATE] != 0.0 || w.st[ST_RETRY] != 0.0 || w.st[ST_GO] != 0.0) return;
    const JDims d = jdims(A.S.N, A.S.Mk[mission]);
    const double nb = d.nblk;
    double tiles = nb * ((nb - 1) * nb / 2 + (nb - 1));  // 64^3 tile products per knot: update tiles + panel tiles of every pass
    if (A.sweep2)  // double passes: two products per update tile outside the pivot block, four per panel block row; an odd order ends with a single pass
        tiles = (double)(d.nblk / 2) * ((nb - 2) * (nb - 1) + 4 * (nb - 2)) + ((d.nblk & 1) ? (nb - 1) * nb / 2 + (nb - 1) : 0.0);
    const double per_knot = tiles * 2.0 * JT * JT * JT;
    w.st[ST_FLOPS] += d.nj * per_knot + 2.0 * (2.0 * d.nj - 1.0) * 2.0 * (double)d.nkp * d.nkp;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------------
JLayout jq_layout(int N, int MS) {
    JLayout L{};
    const JDims d = jdims(N, MS);
    L.N = N, L.MS = MS, L.nkpS = d.nkp, L.nblkS = d.nblk, L.njS = d.nj, L.nch = d.nch;
    L.nred = std::max(N * d.nch, (d.nj * 3 * N + 255) / 256);
    size_t o = 0;
    auto take = [&](size_t n) {
        const size_t r = o;
        o += (n + 31) & ~size_t(31);  // 256-byte granules
        return r;
    };
    const size_t ncp = (size_t)N * d.oq, nrow = (size_t)d.npair * d.oq;
    L.o_state = take(ST_N);
    L.o_segsc = take(MS), L.o_Lk = take(9 * (MS + 1)), L.o_Dk = take(9 * (MS + 1)), L.o_Ek = take(9 * (MS + 1));
    L.o_boxlo = take((size_t)N * MS * 3), L.o_boxhi = take((size_t)N * MS * 3);
    L.o_dxa = take(3 * ncp), L.o_dx = take(3 * ncp);
    L.o_rbase = take((size_t)d.nj * d.nkp), L.o_rhs = take((size_t)d.nj * d.nkp), L.o_wv = take(3 * (size_t)d.nj * d.nkp);  // wv, then rhs0 and dusave of the iterative refinement
    L.o_red = take((size_t)RS_NSLOT * 4 * L.nred);
    L.zero_doubles = o;  // everything up to here is cleared at the start of a run
    for (int p = 0; p < 2; ++p) L.o_bs[p] = take(6 * ncp), L.o_bz[p] = take(6 * ncp), L.o_ps[p] = take(nrow), L.o_pz[p] = take(nrow);
    L.o_pwgt = take(nrow);
    L.o_acc = take((size_t)d.nch * 12 * ncp);
    L.o_Y = take((size_t)4 * d.nblk * JTT), L.o_P = take((size_t)10 * JTT);  // (per chain: two panel tiles per block row; 2 + 2 single pivots, 3 + 3 tiles of double pivots)
    L.o_scr = take((size_t)2 * d.nkp * d.nkp);
    const PolLayout PL = pol_layout(N, MS);
    L.o_inv = take(std::max((size_t)d.nj * d.nkp * d.nkp, PL.big_total));  // (the polish's matrices overlay the inverses: jqp_polish.inc pol_layout)
    L.o_pol = take(PL.total);
    L.o_rhsc = take((size_t)d.nj * d.nkp), L.o_dx2 = take(3 * ncp), L.o_tb = take(6 * ncp), L.o_tp = take(nrow);
    L.stride = o;
    return L;
}

size_t joint_workspace_bytes(int N, int MS) { return jq_layout(N, MS).stride * sizeof(double); }
size_t joint_exchange_bytes(int N, int MS, int K) {  // header + the largest of jq_xfer's three slots (the inverse and its flag)
    const JDims d = jdims(N, MS);
    return (JQ_XHDR + (size_t)K * std::max((size_t)d.nkp * d.nkp + 1, (size_t)(d.nj / 2 + 1) * d.nkp)) * sizeof(double);
}

#define JQ_LAUNCH(kern, grid, lds, ...) hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, __VA_ARGS__)

// One joint QP per mission of the session.  Synchronises the stream once per interior-point iteration (to learn whether any mission is
// still running); everything else is enqueued.
int launch_planner_joint(const DevSession& s, void* ws, hipStream_t st, JointStats* stats, const JointOpts& opts) {
    JArgs A;
    A.S = s, A.ws = (double*)ws, A.L = jq_layout(s.N, s.M);
    // Solver constants (what each one does: jqp.h JArgs).  They are compiled in: the library reads no environment variables
    // (rbp_solver_opts carries the switches a caller may set).  The developer build (-DRBP_DEV_KNOBS, `make dev`) lets experiments override
    // them from the environment (tools/joint_env_sweep.sh).
    A.chain0 = 0, A.fuse_panel = 0;
    A.dreg_mode = 0, A.dreg_scale = 1.0, A.dreg_max = 1e-4, A.ref_step = 0, A.ref_gate = 0, A.retry_only = 0, A.gond_only = 0;
    A.tune[0] = JQ_MU0, A.tune[1] = JQ_SFLOOR, A.tune[2] = 3.0, A.tune[3] = JQ_NBHD_GAMMA, A.tune[4] = JQ_STEP_FRAC;
    A.pol_lh_early = 60, A.pol_lh_final = 160;
    A.gond[0] = 0.3, A.gond[1] = 0.1, A.gond[2] = 0.9;
    // (qp.hip tries at 1e-6 and 1e-8; here the candidate set of a mu = 1e-6 iterate is thousands of rows away from the active set -- the
    // attempt costs more than an iteration and never succeeded on the 50 maps --, so ONE early attempt at 1e-8: 3.07 -> 2.64 s per sweep)
    A.early_mu[0] = 1e-8, A.early_mu[1] = 0.0;
    A.pol_tau = 2e-7, A.pol_vtol = 5e-8, A.pol_adtau = 1, A.exit_mu = 1e-9;
    int pol_wait = 8;
    int sched = opts.schedule, gondzio = opts.corrector ? 1 : 0;
    bool trace = false;
#ifdef RBP_DEV_KNOBS
    {
        const char* e = getenv("RBP_JQ_DREG");  // "mode,scale,max"
        if (e) sscanf(e, "%d,%lf,%lf", &A.dreg_mode, &A.dreg_scale, &A.dreg_max);
        if ((e = getenv("RBP_JQ_LH"))) sscanf(e, "%lf,%lf", &A.pol_lh_early, &A.pol_lh_final);  // round caps of the polish's Lawson-Hanson fallback
        if ((e = getenv("RBP_JQ_TUNE"))) sscanf(e, "%lf,%lf,%lf,%lf,%lf", &A.tune[0], &A.tune[1], &A.tune[2], &A.tune[3], &A.tune[4]);
        if ((e = getenv("RBP_JQ_GOND"))) sscanf(e, "%lf,%lf,%lf", &A.gond[0], &A.gond[1], &A.gond[2]);
        if ((e = getenv("RBP_JQ_EARLY"))) sscanf(e, "%lf,%lf", &A.early_mu[0], &A.early_mu[1]);
        if ((e = getenv("RBP_JQ_POLTAU"))) A.pol_tau = atof(e);
        if ((e = getenv("RBP_JQ_VTOL"))) A.pol_vtol = atof(e);
        if ((e = getenv("RBP_JQ_ADTAU"))) A.pol_adtau = atoi(e);
        if ((e = getenv("RBP_JQ_EXITMU"))) A.exit_mu = atof(e);
        if ((e = getenv("RBP_JQ_POLWAIT"))) pol_wait = atoi(e);
        if ((e = getenv("RBP_JQ_SCHED"))) sched = e[0] == 'b' ? 2 : 1;
        if ((e = getenv("RBP_JQ_GONDZIO"))) gondzio = atoi(e) != 0;
        trace = getenv("RBP_JOINT_TRACE") != nullptr;
    }
#endif
    const JLayout& L = A.L;
    const int K = s.K, N = s.N;
    const JDims dm = jdims(N, s.M);  // the session's largest mission
    const int nsw = N * dm.nch, npost = (dm.nj * 3 * N + 255) / 256, nblk = dm.nblk, ntri = nblk * (nblk + 1) / 2;
    const int mid_max = dm.nj / 2, steps = std::max(mid_max, dm.nj - 1 - mid_max);
    const int nxblk = (N * 3 * dm.oq + 255) / 256, nprep = (9 * N * N + 255) / 256;
    for (int k = 0; k < K; ++k)
        if (hipMemsetAsync(A.ws + (size_t)k * L.stride, 0, L.zero_doubles * sizeof(double), st) != hipSuccess) return RBP_ERR_HIP;
    JQ_LAUNCH(jq_setup, dim3(K), 0, A);
    JQ_LAUNCH(jq_sweep<PASS_INIT>, dim3(nsw, K), 0, A);
    JQ_LAUNCH(jq_ctrl, dim3(K), 0, A, 0, 0);
    std::vector<double> state((size_t)K);
    struct Pinned {  // the poll buffer (freed on every exit path)
        double* p = nullptr;
        ~Pinned() {
            if (p) (void)hipHostFree(p);
        }
    } pinned;
    if (hipHostMalloc((void**)&pinned.p, sizeof(double) * ((size_t)K * ST_N + 2)) != hipSuccess) return RBP_ERR_HIP;
    double* state_h = pinned.p;
    double* xerr_h = pinned.p + (size_t)K * ST_N;  // (stream-ordered pair: JointShard::xerr as of the last poll)
    xerr_h[0] = xerr_h[1] = 0.0;
    // jq_mv keeps one vector of nkp doubles in dynamic LDS: above the default 64 KB the limit has to be raised, above the CU's 160 KB
    // (N > 2275 agents) the launch cannot be made at all
    if ((size_t)dm.nkp * sizeof(double) > 160 * 1024) return RBP_ERR_BAD_ARGUMENT;
    if (hipFuncSetAttribute((const void*)jq_mv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(dm.nkp * sizeof(double))) != hipSuccess) return RBP_ERR_HIP;
    int nref_round = 0;  // refinement steps per solve in this round (the largest any mission asked for; the kernels gate per mission)
    // two-rank factorisation: this rank launches its own chain only (grid.y = 1, A.chain0 = rank) and trades the three pieces of jq_xfer
    const JointShard* sh = opts.shard && opts.shard->nranks == 2 ? opts.shard : nullptr;
    const int ychains = sh ? 1 : 2, my_chain = sh ? sh->rank : 0;
    int xrc = RBP_OK;
    if (sh && ((!sh->exchange && !sh->exchange_stream) || !sh->send || !sh->recv || sh->cap < joint_exchange_bytes(N, s.M, K) || sh->rank < 0 || sh->rank > 1 ||
               (sh->exchange_stream && !sh->xerr)))
        return RBP_ERR_BAD_ARGUMENT;
    if (sh && sh->exchange_stream && hipMemsetAsync(sh->xerr, 0, 2 * sizeof(double), st) != hipSuccess) return RBP_ERR_HIP;
    // the per-round wait of a stream-ordered pair: a peer that is gone leaves the stream blocked in its receive for ever, so the wait polls
    // with a clock; on time-out the hook's owner is asked to bring the exchange down (abort_peer) and the run ends with RBP_ERR_EXCHANGE
    auto wait_round = [&]() -> int {
        if (!(sh && sh->exchange_stream)) return hipStreamSynchronize(st) == hipSuccess ? RBP_OK : RBP_ERR_HIP;
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spin = 0;; ++spin) {
            const hipError_t q = hipStreamQuery(st);
            if (q == hipSuccess) return RBP_OK;
            if (q != hipErrorNotReady) return RBP_ERR_HIP;
            if (sh->timeout_s > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > sh->timeout_s) {
                if (sh->abort_peer) (void)sh->abort_peer(sh->user);
                return rbp_set_error(RBP_ERR_EXCHANGE, "joint QP: timed out waiting for a round of the stream-ordered exchange (the peer rank is gone or stuck); both ranks must abort");
            }
            if (spin > 4000) std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
    };
    auto xerr_check = [&](const double* xe) -> int {  // xe: host copy of sh->xerr taken with the poll
        if (xe[0] == 0.0) return RBP_OK;
        static const char* const names[6] = {"magic", "sequence number", "kind", "byte count", "state hash", "poison word (the peer rank failed)"};
        char msg[320];
        const int f = (int)xe[0] - 1;
        snprintf(msg, sizeof(msg), "joint QP: exchange %.0f does not match the peer rank's: %s differs -- the two ranks of the pair have diverged or one has failed; "
                 "both must abort", xe[1], names[f >= 0 && f < 6 ? f : 0]);
        return rbp_set_error(RBP_ERR_EXCHANGE, msg);
    };
    double xseq = 0;
    auto state_hash = [&]() {  // FNV-1a over the state words polled last and the host variables the launch pattern depends on
        unsigned long long h = 1469598103934665603ull;
        auto mix = [&](const void* p, size_t n) {
            const unsigned char* c = (const unsigned char*)p;
            for (size_t i = 0; i < n; ++i) h = (h ^ c[i]) * 1099511628211ull;
        };
        mix(state_h, sizeof(double) * (size_t)K * ST_N);
        mix(&nref_round, sizeof(nref_round));
        mix(&A.ref_gate, sizeof(A.ref_gate)), mix(&A.gond_only, sizeof(A.gond_only));
        return (double)(h >> 11);  // (53 bits: exact in a double)
    };
    // a rank that cannot go on (HIP error on its side) tells its peer before it returns: one header-only exchange with the poison word
    // set.  If the hook itself is what failed there is nobody to tell through it: the hook's owner must bring the peer down
    // (rbp_rccl_exchange aborts its communicator), both ranks must abort together (include/rbp.h).
    auto xslot = [&](int what) { return what == 0 ? (size_t)dm.nkp * dm.nkp + 1 : what == 1 ? (size_t)dm.nkp : (size_t)(dm.nj / 2) * dm.nkp; };
    // (send / recv pairs must match in size: the poisoned message has the size of the exchange the peer is about to make, `what`)
    auto poison_peer = [&](int what) {
        const size_t bytes = (JQ_XHDR + (size_t)K * xslot(what)) * sizeof(double);
        double h[JQ_XHDR] = {JQ_XMAGIC, xseq, (double)what, (double)bytes, 0.0, 1.0, 0.0, 0.0};
        if (hipMemcpy(sh->send, h, sizeof(h), hipMemcpyHostToDevice) == hipSuccess) (void)sh->exchange(sh->user, sh->send, sh->recv, bytes);
    };
    auto exchange = [&](int what) {
        if (xrc != RBP_OK) return;
        const size_t slot = xslot(what);
        if (slot == 0) return;
        const size_t bytes = (JQ_XHDR + (size_t)K * slot) * sizeof(double);
        const dim3 grid((unsigned)std::min<size_t>((slot + 255) / 256, 2048), K);
        const double mine[JQ_XHDR] = {JQ_XMAGIC, xseq, (double)what, (double)bytes, state_hash(), 0.0, 0.0, 0.0};
        if (sh->exchange_stream) {  // stream-ordered: pack, header, exchange, check, unpack -- all enqueued, nothing waited for
            JQ_LAUNCH(jq_xfer, grid, 0, A, what, 0, my_chain, sh->send + JQ_XHDR, (const double*)nullptr);
            hipLaunchKernelGGL(jq_xhdr_write, dim3(1), dim3(64), 0, st, sh->send, mine[1], mine[2], mine[3], mine[4]);
            if (sh->exchange_stream(sh->user, sh->send, sh->recv, bytes, (void*)st) != 0) {
                xrc = rbp_set_error(RBP_ERR_EXCHANGE, "joint QP: the stream-ordered exchange hook of rbp_session_shard_joint_stream reported a failure");
                return;
            }
            hipLaunchKernelGGL(jq_xhdr_check, dim3(1), dim3(64), 0, st, sh->recv, mine[1], mine[2], mine[3], mine[4], sh->xerr);
            JQ_LAUNCH(jq_xfer, grid, 0, A, what, 1, 1 - my_chain, sh->recv + JQ_XHDR, (const double*)sh->xerr);
            xseq += 1;
            return;
        }
        JQ_LAUNCH(jq_xfer, grid, 0, A, what, 0, my_chain, sh->send + JQ_XHDR, (const double*)nullptr);
        if (hipMemcpyAsync(sh->send, mine, sizeof(mine), hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
            xrc = RBP_ERR_HIP;
            poison_peer(what);
            return;
        }
        if (sh->exchange(sh->user, sh->send, sh->recv, bytes) != 0) {
            xrc = RBP_ERR_EXCHANGE;
            (void)rbp_set_error(RBP_ERR_EXCHANGE, "joint QP: the exchange hook of rbp_session_shard_joint reported a failure");
            return;
        }
        double peer[JQ_XHDR];
        if (hipMemcpy(peer, sh->recv, sizeof(peer), hipMemcpyDeviceToHost) != hipSuccess) {
            xrc = RBP_ERR_HIP;
            return;
        }
        static const char* const names[6] = {"magic", "sequence number", "kind", "byte count", "state hash", "poison word (the peer rank failed)"};
        const int order[6] = {0, 5, 1, 2, 3, 4};  // (a poisoned header is reported as such, whatever else it carries)
        for (int oi = 0; oi < 6; ++oi) {
            const int f = order[oi];
            if (peer[f] == mine[f]) continue;
            char msg[320];
            snprintf(msg, sizeof(msg), "joint QP: exchange %.0f (kind %d) does not match the peer rank's: %s differs (mine %.17g, the peer's %.17g) -- the two ranks of "
                     "the pair have diverged or one has failed; both must abort", xseq, what, names[f], mine[f], peer[f]);
            xrc = rbp_set_error(RBP_ERR_EXCHANGE, msg);
            return;
        }
        xseq += 1;
        JQ_LAUNCH(jq_xfer, grid, 0, A, what, 1, 1 - my_chain, sh->recv + JQ_XHDR, (const double*)nullptr);
    };
    auto substitute = [&](int which_out) {
        A.chain0 = my_chain;
        for (int sidx = 0; sidx < steps; ++sidx) JQ_LAUNCH(jq_mv, dim3(dm.nkp / 16, ychains, K), dm.nkp * sizeof(double), A, 0, sidx);
        A.chain0 = 0;
        if (sh) exchange(1);
        JQ_LAUNCH(jq_mv, dim3(dm.nkp / 16, 1, K), dm.nkp * sizeof(double), A, 1, 0);
        JQ_LAUNCH(jq_mv, dim3(dm.nkp / 16, 1, K), dm.nkp * sizeof(double), A, 3, 0);
        A.chain0 = my_chain;
        for (int sidx = steps - 1; sidx >= 0; --sidx) JQ_LAUNCH(jq_mv, dim3(dm.nkp / 16, ychains, K), dm.nkp * sizeof(double), A, 2, sidx);
        A.chain0 = 0;
        if (sh) exchange(2);
    };
    auto solve = [&](int which_out) {
        if (nref_round > 0) JQ_LAUNCH(jq_refine, dim3(npost, K), 0, A, 0, which_out);
        substitute(which_out);
        JQ_LAUNCH(jq_apply_F, dim3(npost, K), 0, A, which_out);
        for (int rs = 0; rs < nref_round; ++rs) {
            A.ref_step = rs;
            if (which_out == 0)
                JQ_LAUNCH(jq_sweep<PASS_KMUL_A>, dim3(nsw, K), 0, A);
            else if (which_out == 1)
                JQ_LAUNCH(jq_sweep<PASS_KMUL_D>, dim3(nsw, K), 0, A);
            else
                JQ_LAUNCH(jq_sweep<PASS_KMUL_G>, dim3(nsw, K), 0, A);
            JQ_LAUNCH(jq_refine, dim3(npost, K), 0, A, 1, which_out);
            A.ref_gate = rs + 1;
            substitute(which_out);
            A.ref_gate = 0;
            JQ_LAUNCH(jq_refine, dim3(npost, K), 0, A, 2, which_out);
            JQ_LAUNCH(jq_apply_F, dim3(npost, K), 0, A, which_out);
        }
        A.ref_step = 0;
    };
    // two schedules of the sweep: look-ahead (the workgroup that updates the next pivot tile inverts it: one dependent launch less per
    // step -- a lone mission is bound by that chain) / bulk (thousands of tiles per launch: a leaner update kernel at three workgroups per
    // CU, the pivot inverse in a launch of its own)
    // (round 5: since the look-ahead tile is taken by the first workgroup of its chain and jq_update runs three workgroups per CU, the
    // look-ahead schedule is ahead at every resident-set size measured -- 64 agents: 50 missions 1721 against 1656, 200 missions 2278 against
    // 2227 agent-trajectories/s, profiles/r05_joint_lookfirst_ab.txt -- so the automatic choice is look-ahead; bulk stays selectable)
    const bool bulk = sched ? sched >= 2 : false;
    // ... schedule 3: the bulk schedule with TWO pivot tiles per pass over the matrix (jq_pivot2 / jq_panel2 / jq_update2_bulk: the update is
    // bound by HBM, and a pass reads and writes the whole lower triangle).  Opt-in: measured at 64 agents (9 tiles per knot) the update
    // kernels' time falls by 28 % but the pivot block's serial chain and the panel's doubled arithmetic take most of it back (+3.5 % at 200
    // resident missions, nothing at 50, a lone mission LOSES 25-30 %): DESIGN.md 3.5
    A.sweep2 = sched == 3 && nblk >= 2 ? 1 : 0;
    const size_t lds_pivot2 = (size_t)3 * JT * LDA * sizeof(double);
    // jq_update takes its LDS dynamically (52.7 KB: three workgroups per CU).  Few tiles per launch (one chain of a lone mission: the launch is
    // as long as the workgroup that carries the look-ahead inversion) run better with TWO workgroups per CU -- asked for by padding the
    // request (measured, 256 agents: one chain per launch 3.76 s against 4.00 s; two chains, 1332 tiles, 4.40 against 4.27 the other way)
    const size_t lds_update_min = (size_t)JT * LDA * sizeof(double) + sizeof(InvScratch) + JQ_UPDATE_LDS_EXTRA;
    auto few_tiles = [&](int nchain) { return (size_t)K * nchain * ntri < 1024; };  // (about a round of workgroups per launch)
    auto lds_update_for = [&](int nchain) { return few_tiles(nchain) ? (size_t)72 * 1024 : lds_update_min; };
    const size_t lds_update = lds_update_min;  // (the polish's S_AA sweeps: one chain, usually few tiles -- but several missions at once)
    if (hipFuncSetAttribute((const void*)jq_update, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(72 * 1024)) != hipSuccess) return RBP_ERR_HIP;
    if (A.sweep2 && hipFuncSetAttribute((const void*)jq_pivot2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pivot2) != hipSuccess) return RBP_ERR_HIP;
    auto factor_knot = [&](int sidx, int mid) {
        const int nchain = mid ? 1 : ychains;
        A.chain0 = mid ? 0 : my_chain;
        struct Reset {
            int& c;
            ~Reset() { c = 0; }
        } reset{A.chain0};
        JQ_LAUNCH(jq_prep, dim3(nprep, nchain, K), 0, A, sidx, mid);
        if (A.sweep2) {
            for (int k = 0; k + 1 < nblk; k += 2) {
                JQ_LAUNCH(jq_pivot2, dim3(1, nchain, K), lds_pivot2, A, sidx, mid, k);
                if (nblk > 2) JQ_LAUNCH(jq_panel2, dim3(nblk, nchain, K), 0, A, sidx, mid, k);
                JQ_LAUNCH(jq_update2_bulk, dim3(ntri, nchain, K), 0, A, sidx, mid, k);
            }
            if (nblk & 1) {
                const int k = nblk - 1;
                JQ_LAUNCH(jq_pivot0, dim3(1, nchain, K), 0, A, 0, sidx, mid, k);
                JQ_LAUNCH(jq_panel, dim3(nblk, nchain, K), 0, A, 0, sidx, mid, k);
                JQ_LAUNCH(jq_update_bulk, dim3(ntri, nchain, K), 0, A, 0, sidx, mid, k);
            }
            return;
        }
        JQ_LAUNCH(jq_pivot0, dim3(1, nchain, K), 0, A, 0, sidx, mid, 0);
        for (int k = 0; k < nblk; ++k) {
            if (bulk && k > 0) JQ_LAUNCH(jq_pivot0, dim3(1, nchain, K), 0, A, 0, sidx, mid, k);
            A.fuse_panel = !bulk && few_tiles(nchain) ? 1 : 0;
            if (nblk > 1 && !A.fuse_panel) JQ_LAUNCH(jq_panel, dim3(nblk, nchain, K), 0, A, 0, sidx, mid, k);
            if (bulk)
                JQ_LAUNCH(jq_update_bulk, dim3(ntri, nchain, K), 0, A, 0, sidx, mid, k);
            else
                JQ_LAUNCH(jq_update, dim3(ntri, nchain, K), lds_update_for(nchain), A, 0, sidx, mid, k);
        }
    };
    A.trace = trace ? 1 : 0;
    // ---- active-set polish of the missions whose control kernel asked for it (jqp_polish.inc); host-driven state machine, one
    // synchronisation per stage
    const PolLayout PL = pol_layout(N, s.M);
    const int ncmax = pol_ncmax(N);
    const size_t nrows_all = 6 * (size_t)dm.ncp + (size_t)dm.npair * dm.oq;
    std::vector<int> cnt_h((size_t)K * PC_N);
    int polish_rounds = 0;
    (void)hipFuncSetAttribute((const void*)jp_z, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(ncmax * sizeof(double)));
    auto poll = [&]() -> bool {
        return hipMemcpy2DAsync(state_h, sizeof(double) * ST_N, A.ws + L.o_state, L.stride * sizeof(double), sizeof(double) * ST_N, K,
                                hipMemcpyDeviceToHost, st) == hipSuccess &&
               hipMemcpy2DAsync(cnt_h.data(), sizeof(int) * PC_N, A.ws + L.o_pol + PL.cnt, L.stride * sizeof(double), sizeof(int) * PC_N, K,
                                hipMemcpyDeviceToHost, st) == hipSuccess &&
               wait_round() == RBP_OK;
    };
    auto polish = [&]() -> int {
        for (int k = 0; k < K; ++k)
            if (hipMemsetAsync(A.ws + (size_t)k * L.stride + L.o_pol + PL.pos, 0xFF, nrows_all * sizeof(int), st) != hipSuccess) return RBP_ERR_HIP;
        JQ_LAUNCH(jp_begin, dim3(K), 18 * dm.nj * sizeof(double), A);
        JQ_LAUNCH(jp_c0, dim3(npost, K), 0, A);
        JQ_LAUNCH(jp_chain_mv, dim3((dm.nj * 9 * N + 255) / 256, K), 0, A, 0);
        JQ_LAUNCH(jq_sweep<PASS_CAND>, dim3(nsw, K), 0, A);
        for (int guard = 0; guard < 4000; ++guard) {
            if (!poll()) return RBP_ERR_HIP;
            bool any_new = false, any_bpp = false, any_primal = false, any_refine = false;
            int nc_max = 0, nblk_max = 0;
            for (int k = 0; k < K; ++k) {
                const double* q = state_h + (size_t)k * ST_N;
                const int* c = cnt_h.data() + (size_t)k * PC_N;
                if (q[ST_STATE] != 0.0 || q[ST_GO] == 0.0) continue;
                if (q[ST_PSTATE] == (double)PS_SOLVE) {
                    if (c[PC_NRAW] > 0)
                        any_new = true, nc_max = std::max(nc_max, std::min(ncmax, c[PC_NCAND] + c[PC_NRAW]));
                    else if (!c[PC_BPPDONE])
                        any_bpp = true, nblk_max = std::max(nblk_max, c[PC_NBLK]);
                } else if (q[ST_PSTATE] == (double)PS_PRIMAL) {
                    any_primal = true, any_refine = any_refine || c[PC_REFINE];
                }
                if (trace)
                    fprintf(stderr, "[jqp]   polish mission %d pstate %.0f ncand %d nraw %d nA %d ninf %d bppit %d done %d round %d outer %d reason %d\n", k,
                            q[ST_PSTATE], c[PC_NCAND], c[PC_NRAW], c[PC_NA], c[PC_NINF], c[PC_BPPIT], c[PC_BPPDONE], c[PC_ROUND], c[PC_OUTER], c[PC_REASON]);
            }
            polish_rounds++;
            if (any_new) {
                JQ_LAUNCH(jp_sort, dim3((ncmax + 255) / 256, K), 0, A);
                JQ_LAUNCH(jp_sort_fin, dim3(K), 0, A);
                JQ_LAUNCH(jp_S, dim3((unsigned)(((size_t)nc_max * nc_max + 255) / 256), K), 0, A);
                JQ_LAUNCH(jp_twin, dim3((nc_max + 255) / 256, K), 0, A);
                JQ_LAUNCH(jp_bpp, dim3(K), 0, A, 0);
            } else if (any_bpp) {
                if (nblk_max > 0) {
                    JQ_LAUNCH(jp_gather, dim3(nblk_max * nblk_max, K), 0, A);
                    JQ_LAUNCH(jq_pivot0, dim3(1, 1, K), 0, A, 1, 0, 0, 0);
                    for (int k = 0; k < nblk_max; ++k) {
                        A.fuse_panel = (size_t)K * nblk_max * (nblk_max + 1) / 2 < 1024 ? 1 : 0;
                        if (nblk_max > 1 && !A.fuse_panel) JQ_LAUNCH(jq_panel, dim3(nblk_max, 1, K), 0, A, 1, 0, 0, k);
                        JQ_LAUNCH(jq_update, dim3(nblk_max * (nblk_max + 1) / 2, 1, K), lds_update, A, 1, 0, 0, k);
                    }
                    JQ_LAUNCH(jp_z, dim3(nblk_max * 4, K), nblk_max * JT * sizeof(double), A, 0);
                    for (int rr = 0; rr < 2; ++rr) {  // two refinement steps: the multipliers' signs drive the exchange
                        JQ_LAUNCH(jp_z, dim3(nblk_max * 4, K), nblk_max * JT * sizeof(double), A, 1);
                        JQ_LAUNCH(jp_z, dim3(nblk_max * 4, K), nblk_max * JT * sizeof(double), A, 2);
                    }
                }
                JQ_LAUNCH(jp_g, dim3((ncmax + 3) / 4, K), 0, A);
                JQ_LAUNCH(jp_bpp, dim3(K), 0, A, 1);
            } else if (any_primal) {
                if (any_refine) {
                    int nb = 0;
                    for (int k = 0; k < K; ++k) nb = std::max(nb, cnt_h[(size_t)k * PC_N + PC_NBLK]);
                    if (nb > 0) JQ_LAUNCH(jp_z, dim3(nb * 4, K), nb * JT * sizeof(double), A, 3);
                }
                JQ_LAUNCH(jp_jz, dim3(npost, K), 0, A);
                JQ_LAUNCH(jp_chain_mv, dim3((dm.nj * 9 * N + 255) / 256, K), 0, A, 1);
                JQ_LAUNCH(jp_dx, dim3(npost, K), 0, A);
                JQ_LAUNCH(jq_sweep<PASS_VERIFY>, dim3(nsw, K), 0, A);
                JQ_LAUNCH(jp_check, dim3(K), 0, A);
            } else
                break;
        }
        JQ_LAUNCH(jp_apply, dim3(nxblk, K), 0, A, 0);
        JQ_LAUNCH(jp_apply, dim3((ncmax + 255) / 256, K), 0, A, 1);
        JQ_LAUNCH(jp_end, dim3(K), 0, A);
        return RBP_OK;
    };
    int iters = 0, rc = RBP_OK, go_waited = 0;
    const int max_rounds = JQ_MAX_ITERS + 48;
    for (int it = 0; it < max_rounds; ++it) {
        if (it == 0) JQ_LAUNCH(jq_sweep<PASS_BUILD>, dim3(nsw, K), 0, A);
        JQ_LAUNCH(jq_post<0>, dim3(npost, K), 0, A);
        JQ_LAUNCH(jq_ctrl, dim3(K), 0, A, 1, (int)(it == 0));
        JQ_LAUNCH(jq_unstep, dim3(nxblk, K), 0, A);
        hipLaunchKernelGGL(jq_unstep_done, dim3(K), dim3(64), 0, st, A);
        // is any mission still running?  (one synchronisation per iteration)
        if (hipMemcpy2DAsync(state_h, sizeof(double) * ST_N, A.ws + L.o_state, L.stride * sizeof(double), sizeof(double) * ST_N, K,
                             hipMemcpyDeviceToHost, st) != hipSuccess ||
            (sh && sh->exchange_stream && hipMemcpyAsync(xerr_h, sh->xerr, 2 * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess)) {
            rc = RBP_ERR_HIP;
            break;
        }
        if ((rc = wait_round()) != RBP_OK) break;
        if ((rc = xerr_check(xerr_h)) != RBP_OK) break;  // (a stream-ordered exchange of the last round did not match the peer's)
        bool running = false;
        for (int k = 0; k < K; ++k) running = running || state_h[(size_t)k * ST_N + ST_STATE] == 0.0;
        if (trace)
            for (int k = 0; k < std::min(K, 4); ++k) {
                const double* q = state_h + (size_t)k * ST_N;
                fprintf(stderr, "[jqp] round %d mission %d state %.0f iter %.0f retry %.0f mu %.3e pres %.3e dres %.3e sigmu %.3e aaff %.3f alpha %.4f bt %.0f\n", it, k,
                        q[ST_STATE], q[ST_ITER], q[ST_RETRY], q[ST_MU], q[ST_PRES], q[ST_DRES], q[ST_SIGMU], q[ST_AAFF], q[ST_ALPHA], q[ST_BT]);
            }
        if (!running) break;
        nref_round = 0;
        for (int k = 0; k < K; ++k)
            if (state_h[(size_t)k * ST_N + ST_STATE] == 0.0) nref_round = std::max(nref_round, (int)state_h[(size_t)k * ST_N + ST_NREF]);
        bool any_go = false, any_ipm = false;
        for (int k = 0; k < K; ++k) {
            const bool run_k = state_h[(size_t)k * ST_N + ST_STATE] == 0.0, go_k = state_h[(size_t)k * ST_N + ST_GO] != 0.0;
            any_go = any_go || (run_k && go_k), any_ipm = any_ipm || (run_k && !go_k);
        }
        // A polish call costs as many exchange rounds as its SLOWEST mission needs, whoever else takes part, and with many missions the
        // requests trickle in over a dozen interior-point rounds: requests wait up to pol_wait rounds for company while other missions still
        // iterate (a waiting mission is frozen -- every interior-point kernel skips missions with ST_GO set -- and loses nothing: the round
        // runs anyway); a session whose missions are all waiting, or a lone mission, is served at once.
        if (any_go && any_ipm && go_waited < pol_wait) {
            go_waited++;
            any_go = false;
        }
        if (any_go) {
            go_waited = 0;
            if ((rc = polish()) != RBP_OK) break;
            // (missions the polish has finished are skipped by the kernels below; if none is left the next poll ends the loop)
        }
        iters++;
        bool all_retry = true;
        for (int k = 0; k < K; ++k)
            if (state_h[(size_t)k * ST_N + ST_STATE] == 0.0 && state_h[(size_t)k * ST_N + ST_RETRY] == 0.0) all_retry = false;
        if (all_retry) {  // every running mission only repeats its update sweep
            A.retry_only = 1;
            for (int rep = 0; rep < 8; ++rep) {
                JQ_LAUNCH(jq_stepx, dim3(nxblk, K), 0, A);
                JQ_LAUNCH(jq_sweep<PASS_UPBUILD>, dim3(nsw, K), 0, A);
                JQ_LAUNCH(jq_ctrl, dim3(K), 0, A, 4, 0);
            }
            A.retry_only = 0;
            continue;
        }
        JQ_LAUNCH(jq_count, dim3(K), 0, A);
        for (int sidx = 0; sidx < steps; ++sidx) factor_knot(sidx, 0);
        if (sh) exchange(0);
        factor_knot(0, 1);
        if (hipPeekAtLastError() != hipSuccess) {  // (a launch the device refuses must not pass for a QP that does not converge)
            if (sh && xrc == RBP_OK) poison_peer(1);  // (the peer's next exchange is the forward vector of solve(0))
            return RBP_ERR_HIP;
        }
        if (xrc != RBP_OK) return xrc;  // (the peer rank is gone or the exchange hook failed: nothing sensible can follow)
        solve(0);
        JQ_LAUNCH(jq_sweep<PASS_AFF>, dim3(nsw, K), 0, A);
        JQ_LAUNCH(jq_ctrl, dim3(K), 0, A, 2, 0);
        JQ_LAUNCH(jq_post<1>, dim3(npost, K), 0, A);
        solve(1);
        JQ_LAUNCH(jq_sweep<PASS_STEP>, dim3(nsw, K), 0, A);
        JQ_LAUNCH(jq_ctrl, dim3(K), 0, A, 3, gondzio);
        if (gondzio) {
            // one centrality corrector on the factorisation already paid for: a sweep for the rows' target shifts, a solve, a sweep for
            // the new step length (missions whose Mehrotra step is long enough skip all of it)
            A.gond_only = 1;
            JQ_LAUNCH(jq_sweep<PASS_GOND>, dim3(nsw, K), 0, A);
            JQ_LAUNCH(jq_post<2>, dim3(npost, K), 0, A);
            solve(2);
            JQ_LAUNCH(jq_sweep<PASS_STEPG>, dim3(nsw, K), 0, A);
            JQ_LAUNCH(jq_ctrl, dim3(K), 0, A, 5, 0);
            JQ_LAUNCH(jq_gcopy, dim3(nxblk, K), 0, A);
            A.gond_only = 0;
        }
        JQ_LAUNCH(jq_stepx, dim3(nxblk, K), 0, A);
        JQ_LAUNCH(jq_sweep<PASS_UPBUILD>, dim3(nsw, K), 0, A);
        JQ_LAUNCH(jq_ctrl, dim3(K), 0, A, 4, 0);
        // a step the wide-neighbourhood test refused is repeated with 0.8 alpha right away (three launches that do nothing for the
        // missions whose step was accepted), not a round later: a round is ~600 launches
        A.retry_only = 1;
        for (int rep = 0; rep < 6; ++rep) {
            JQ_LAUNCH(jq_stepx, dim3(nxblk, K), 0, A);
            JQ_LAUNCH(jq_sweep<PASS_UPBUILD>, dim3(nsw, K), 0, A);
            JQ_LAUNCH(jq_ctrl, dim3(K), 0, A, 4, 0);
        }
        A.retry_only = 0;
        if (xrc != RBP_OK) return xrc;
    }
    JQ_LAUNCH(jq_finish, dim3(K), 0, A);
    if (stats) stats->rounds = iters, stats->polish_rounds = polish_rounds;
    if (hipGetLastError() != hipSuccess) rc = RBP_ERR_HIP;
    return rc;
}
