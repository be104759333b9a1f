// qp.hip — the batched RBP QP on gfx950: assembly, interior-point solve, epilogue.
//
// Replaces RBPPlanner::update (reference: swarm_planner/include/rbp_planner.hpp:33-84): buildConstMtx :100-109,
// populatebyrow :551-688, cplex.solve() :158, the Bernstein->monomial loop :167-196, timeScale :209-266.
//
// Formulation (DESIGN.md "QP"): the equality rows of Aeq_base (:353-405) are C2 continuity at the knots plus the
// pinned end states, so per (agent, dim) and interior knot j the three control points right of the knot are free
// (u_j) and the three left of it are L_j u_j.  In these coordinates the batch QP has NO equality rows, every
// inequality row (SFC bound :626-635, RSFC :638-684) touches exactly one knot, and the Newton matrix of a
// primal-dual interior-point method, F'(2Q)F + J' W J, is block tridiagonal over the M-1 knots with dense
// blocks of order nk = 9 * (batch agents).  One workgroup runs one mission's whole batch schedule in one launch:
//   * three row sweeps per interior-point iteration stream the row state -- (s, z) only, in sliced-ELLPACK order: one coalesced
//     512-byte access per wavefront and array, see QpWs -- from HBM (affine sweep incl. both parts of the corrector rhs; step
//     sweep; step into the ping-pong arrays + neighbourhood test + next iteration's weights);
//   * per-control-point 3x3 accumulators are expanded into the knot blocks (no atomics) by the six waves that would otherwise
//     idle behind the factorisation chains, block by block just ahead of them (twisted_factor);
//   * the block-tridiagonal factorisation: nk <= 36 twisted two-wave chains, L D L' with column images broadcast through LDS
//     (knot_lds.inc), the explicit inverse factor M = L^-T by a companion wave, MFMA rank-k updates; wider batches an MFMA-tiled path
//     (LDS-resident for nk <= 72); substitutions as matrix-vector products with the staged M;
//   * an active-set polish (qp_polish.inc) turns the interior-point answer into the exact optimum, verified by a full
//     KKT check; from the second Gauss-Seidel pass on it is tried before any interior-point iteration.
// Batches of a mission are solved strictly in the reference's order (Gauss-Seidel, :140-148); parallelism comes
// from the 3*nb coupled blocks inside a batch and from the K missions of a session.  The file is compiled twice
// (512 / 256 threads per workgroup, see the note above planner_workspace_bytes).
#include <algorithm>
#include <type_traits>
#include <vector>

#include "rbp_dev.h"

// The QP is tolerance-judged floating point: allow FMA contraction here (the Makefile disables it globally because
// corridor.hip must round exactly like the reference's float32 code).
#pragma clang fp contract(fast)

#include "knot_lds.inc"  // (below the pragma: its multiply-adds must contract)

#ifndef QP_THREADS
#define QP_THREADS 512
#endif
// QP_MAX_NB (rbp_dev.h): nk <= 36: wave-register path; wider: MFMA-tiled path (blocks in LDS up to nk = 72, else global)
#define QP_MAX_ITERS 80
#ifndef QP_POLISH_FIRST
#define QP_POLISH_FIRST 1
#endif
#ifndef QP_WARM_TAU
#define QP_WARM_TAU 1e-3  // polish-first (Gauss-Seidel passes >= 2): rows closer than this to their bound are candidates
#endif
#ifndef QP_MU0
#define QP_MU0 3e-1     // interior-point start: z = mu0 / s with s = max(slack, s_floor)  (tuned on the 50-map sweep)
#endif
#ifndef QP_UNI_POLISH
#define QP_UNI_POLISH 1
#endif
#ifndef QP_ROW_UNROLL  // frozen-row stream of a sweep: rows unrolled per thread.  A/B on one box (r03, 2000 missions resident): 1: 82.8 k,
#define QP_ROW_UNROLL 2  // 2: 86.0 k, 4: 78.8 k agent-trajectories/s
#endif
#ifndef QP_ROW_PF
#define QP_ROW_PF 0  // prefetch distance (rows) of the frozen-row stream; 0 = plain loop unrolled QP_ROW_UNROLL times
#endif
#ifndef QP_FUSE_F
#define QP_FUSE_F 2  // (round 6) 1: apply_F fused into the substitutions' epilogue; 2: and rhs_from_acc into their prologue (see SolveOut)
#endif
#ifndef QP_BLK_PRE
#define QP_BLK_PRE 1
#endif
#ifndef QP_BLK_MASK
#define QP_BLK_MASK 0x1EE  // the passes (bit = PASS_*) that stream in blocks: the four sweeps of the interior-point loop (BUILD, AFF, STEP, UPBUILD) and the polish's
                           // (CAND, VERIFY, CAND_GEO)
#endif
#ifndef QP_ROW_BLK
#define QP_ROW_BLK (QP_THREADS >= 512 ? 0 : 4)  // (round 6) frozen-row stream in BLOCKS of this many rows whose loads are issued a block ahead (see row_pass); 0 = off.
                                             // The 256-thread build (missions that share the memory system with 511 others); A/B 3 / 4 / 5 / 6: 4
#endif
// a generic pointer into the workspace as the GLOBAL pointer it is: loads / stores through it are global_* instructions (one memory counter,
// counted in order) instead of flat_* (both counters, and the compiler waits for every outstanding access at the first use of any)
#define QG(p) ((__attribute__((address_space(1))) double*)(p))
#define QGC(p) ((__attribute__((address_space(1))) const double*)(p))
#define QGF(p) ((__attribute__((address_space(1))) const float*)(p))
#ifndef QP_FAR_SLACK
// REDUCED ROW SET of the interior-point phase (round 5).  A frozen-neighbour row whose slack at the batch QP's starting point exceeds this
// many metres for all six control points of its (agent, segment) group is FAR: the interior-point sweeps do not walk it (no (s, z), no
// weight, no ratio test).  Correctness does not depend on the choice: the active-set polish verifies EVERY row of the full QP at the point it
// proposes (far rows that come out violated join the candidates and the dual is solved again), and a batch QP whose polish is refused -- or
// whose interior-point method fails -- while far rows exist is solved again from its starting point with every row near.  Applies to the
// first Gauss-Seidel pass of the sequential schedule with the polish on.  The value travels in rbp_solver_opts.qp_far_slack (DevParam::far_slack);
// this is its default (abi/session.hip rbp_solver_opts_defaults).
#define QP_FAR_SLACK 0.7  // (A/B on one box, 2000 missions resident: off 100.1 k, 1.0 m 123.2 k, 0.7 m 127.7 k, 0.5 m 127.2 k agent-trajectories/s;
                          // batch QPs solved twice on the 50-map sweep: 0 / 1 / 8 of 800; 0.25 m costs iterations: profiles/r05_ab_reduced_rows.txt)
#endif
#ifndef QP_SIGMA_POW
#define QP_SIGMA_POW 3  // Mehrotra's centring exponent
#endif
#ifndef QP_NBHD_GAMMA
#define QP_NBHD_GAMMA 1e-3  // wide neighbourhood: no complementarity product below gamma * mu
#endif
#ifndef QP_NBHD_BACKOFF
#define QP_NBHD_BACKOFF 0.8  // a step the wide-neighbourhood test refuses is repeated with this fraction of its length
#endif
#ifndef QP_STEP_FRAC
#define QP_STEP_FRAC 0.997  // fraction of the step to the boundary (A/B on one box: 0.98 +5 %, 0.99 +1.4 %, 0.997 and 0.999 best, 0.9999 +20 % time)
#endif
#ifndef QP_SFLOOR
#define QP_SFLOOR 1e-1
#endif
#ifndef QP_WAVES_PER_EU
#define QP_WAVES_PER_EU 2  // 256 VGPRs per lane in both builds (one 512-thread or two 256-thread workgroups per CU)
#endif
#ifndef QP_CHAIN_PRIO
#define QP_CHAIN_PRIO 3  // s_setprio level of the waves that run a dependent chain (factor, substitutions, dual active-set solve)
#endif
#ifndef QP_STAGE_BUFS
#ifndef QP_STAGE_BUFS_BIG
#define QP_STAGE_BUFS_BIG 3
#endif
#define QP_STAGE_BUFS (QP_THREADS >= 512 ? QP_STAGE_BUFS_BIG : 2)  // (the 256-thread build shares a CU's LDS between two workgroups)
#endif
#ifndef QP_STAGE_REGS
#define QP_STAGE_REGS (QP_THREADS >= 512 ? 0 : 2)  // register sets of a staging thread whose loads stay in flight across step barriers (solve_staged; 0 = load and store within one step): the 256-thread build, whose missions share the memory system with 511 others
#endif
#ifndef QP_RCP_NEWTON
#define QP_RCP_NEWTON 2
#endif
#ifndef QP_T_WRITE_FULL
#define QP_T_WRITE_FULL 0  // LDS-staged assembly: 1 = whole rows of T_j leave for global memory, 0 = the half the factorisation reads
#endif
#ifndef QP_MID_FOLLOW
#define QP_MID_FOLLOW (QP_THREADS >= 512)  // see wave_factor_mid
#endif
#ifndef QP_EARLY_TRIES
#define QP_EARLY_TRIES 2
#endif
#ifndef QP_EARLY_TOL
#define QP_EARLY_TOL 1e-6
#endif
#ifndef QP_EARLY_POLISH
#define QP_EARLY_POLISH 1  // try the active-set polish before the interior-point loop has fully converged
#endif
// LDS doubles used by polish_qp<36>: 2 blocks + packed factor + vectors + int arrays (see qp_polish.inc)
#include "lh_layout.h"
#define PL_NC 256    // polish: candidate rows
#define PL_PMAX 160  // polish: rows simultaneously active in the dual solve (wide batches; 112 on the wave path, whose
                     // workgroups are meant to share a CU's LDS in pairs)
#define PL_PBIG 192  // second attempt of a polish whose dual solve ran out of capacity: factor in global memory (rare: ~1 batch QP in 800)
__host__ __device__ inline int polish_pmax(int nk) { return nk <= 36 ? 112 : PL_PMAX; }
__host__ __device__ inline size_t polish_ws_doubles(int nj, int nk) {  // cand, V, S, counters, big factor
    return PL_NC * 14 + (size_t)(PL_NC + 1) * nj * nk + PL_NC * PL_NC + 8 + lhp_size(PL_PBIG);
}
__host__ __device__ inline int polish_lds_doubles(int nk) {
    const int pm = polish_pmax(nk);
    return lhp_size(pm) + 3 * PL_NC + 3 * pm + (pm + 2 * PL_NC + 8) / 2 + 8;
}

namespace {

#ifdef QP_PROFILE
#define PROF_DECL long long prof_t0 = wall_clock64(), prof_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define PROF(i)                                \
    do {                                       \
        __syncthreads();                       \
        long long t_ = wall_clock64();         \
        prof_acc[i] += t_ - prof_t0;           \
        prof_t0 = t_;                          \
    } while (0)
#define PROF_FLUSH(scal)                                                         \
    do {                                                                         \
        if (threadIdx.x == 0)                                                    \
            for (int i_ = 0; i_ < 12; ++i_) scal[SC_PROF0 + i_] += (double)prof_acc[i_]; \
    } while (0)
#else
#define PROF_DECL
#define PROF(i)
#define PROF_FLUSH(scal)
#endif

// ------------------------------------------------------------------------------------------------------------
// dimensions of one batch QP
// ------------------------------------------------------------------------------------------------------------
struct QpDims {
    int N, M, oq;       // agents, segments, 6M
    int nb, NF, npb;    // batch agents, frozen agents, in-batch pairs
    int nk, nj, ld;     // block order 9*nb, knots M-1, padded leading dimension
    int ldb;            // leading dimension of the knot blocks in global memory: nk (wave path) or nk rounded up to 16 (tiled path)
    int first;          // first agent of the batch (batches are contiguous: qi / batch_size == l)
    int ncol0;          // rows every control point of a batch agent has: 6 bound rows + (nb - 1) in-batch pair rows
    int ntile;          // 64-lane tiles of control points: ceil(nb * oq / 64)
    size_t nslot_max;   // capacity of the row arrays: ntile * 64 * (ncol0 + NF)
};

__host__ __device__ inline QpDims make_dims(int N, int M, int first, int nb) {
    QpDims d;
    d.N = N, d.M = M, d.oq = 6 * M, d.nb = nb, d.NF = N - nb, d.npb = nb * (nb - 1) / 2;
    d.nk = 9 * nb, d.nj = M - 1, d.ld = d.nk + 1, d.first = first;
    d.ldb = d.nk <= 36 ? d.nk : ((d.nk + 15) & ~15);
    d.ncol0 = 6 + nb - 1, d.ntile = (nb * d.oq + 63) / 64;
    d.nslot_max = (size_t)d.ntile * 64 * (d.ncol0 + d.NF);
    return d;
}

// ROW STORAGE ("sliced ELLPACK", slice = one wavefront).  A sweep hands one control point (agent a, segment, i) of a batch agent
// to a thread; that thread owns ALL rows that touch its control point:
//     6 bound rows  |  nb - 1 in-batch pair rows  |  cnt(a, segment) frozen-neighbour rows that survived the presolve
// Control points are numbered wi = 6 * rank(a, segment) + i with the (a, segment) groups ranked by falling row count, 64
// consecutive wi form a tile (= the lanes of one wavefront), and row `idx` of control point wi lives in slot
//     tile_base[wi / 64] + idx * 64 + (wi % 64)
// so every load/store of a sweep is ONE coalesced 512-byte access per wavefront and array, and consecutive rows of a thread
// are consecutive 512-byte lines: a pure stream.  (The previous layout kept the rows of a group contiguous: a wavefront then
// touched eleven 48-byte pieces of different cache lines per array and step, and HBM-side traffic was ~2.5x the bytes used.)
// A pair row is shared by two control points (one of each agent) and is STORED TWICE, once in each one's column: both copies
// are updated by the same arithmetic on the same inputs, hence stay bit-identical; reductions count the copy of the lower
// agent only.  Only (s, z) are stored per row -- the corrector term and the step (ds, dz) are recomputed where they are needed
// (they are functions of s, z and the two direction vectors), which halves the bytes of a sweep; the step sweep writes the
// new (s, z) into a second pair of arrays (ping-pong) so that a rejected step can be repeated from the old state.
struct QpWs {
    double *s, *z, *s2, *z2;        // row state [nslot]: current / next
    double *rh;                     // [nslot] frozen-neighbour rows: the constant of slack = rh - n . x_a
    double *cc, *ds;                // [nslot] polish only: candidate marks / slack at the trial point
    double *cpacc;                  // [12][nb*oq]: S(6) yv(3) gz(3) per control point, ALL its rows (bounds, pairs, frozen); component-major, so
                                    // that the passes that read three or six of the twelve (rbase, rhs, assembly) fetch only those
    double *pwgt;                   // [npb*oq] Newton weight of every in-batch pair row (off-diagonal blocks of the knot matrices)
    double *dx, *dxa, *cvec;        // [nb*3*oq]
    double *rbase, *rhs;            // [(M-1)*nk]
    double *Td, *To;                // [(M-1)][nk*nk], [(M-2)][nk*nk]
    double *Lf;                     // [(M-1)][2][nk*nk]: L_jj and L_{j+1,j}, both stored [k][r] (element (r,k) at k*nk + r)
    double *boxlo, *boxhi;          // [nb][M][3]
    double *Lk, *Dk, *Ek;           // [M+1][9]
    double *segsc;                  // [M] dt^-5 (build_Q_p :349-351)
    int *flist, *fcnt, *fbase;      // non-redundant frozen neighbours per (batch agent, segment): [nb][M][NF], [nb][M], [nb][M]
    int *fnear;                     // (round 5) [nb][M] how many of a group's neighbours are NEAR (they come first in its rows; the far ones sit at
                                    // the back of its flist): the interior-point sweeps only walk the near rows (see QP_FAR_SLACK)
    int *fperm, *frank;             // [nb][M]: groups ordered by falling row count (fperm[rank] = group, frank[group] = rank)
    int *tile_base;                 // [ntile + 1] first slot of each tile
    int *wi_of;                     // [nb*oq] column (wi) of control point (a, j6)
    float* nrm;                     // [sum cnt][3] signed normal of frozen row (group, idx): the six rows of a group share it
    double* polish;                 // PolishWs storage
    double* prof;                   // QP_PROFILE builds: this mission's diagnostic scalars (chain-side timers), else nullptr
};

__host__ __device__ inline size_t ws_int_count(int N, int M, int nbmax) {
    QpDims d = make_dims(N, M, 0, nbmax);
    return (size_t)nbmax * M * N /*flist*/ + 5 * (size_t)nbmax * M /*fcnt fbase fperm frank fnear*/ + d.ntile + 1 + (size_t)nbmax * d.oq /*wi_of*/ +
           3 * (size_t)nbmax * M * N /*nrm (floats)*/ + 8;
}

// PHASE-SPLIT PATH (round 5, see ph_advance): per mission a state record and the partial reductions of the chip-wide row sweeps live behind
// the workspace proper (offset ws_core_doubles rounded up to four doubles)
#define PH_STATE_DOUBLES 32  // >= sizeof(PhState) / 8
__host__ __device__ inline int ph_nwg(int nb, int M) { return (nb * 6 * M + 255) / 256; }  // workgroups of a chip-wide sweep over one batch QP
__host__ __device__ inline size_t ph_extra_doubles(int M, int nbmax) { return PH_STATE_DOUBLES + 8 * (size_t)ph_nwg(nbmax, M) + 8; }
__host__ __device__ inline size_t ws_core_doubles(int N, int M, int nbmax);
__host__ __device__ inline size_t ph_state_offset(int N, int M, int nbmax) { return (ws_core_doubles(N, M, nbmax) + 3) & ~size_t(3); }
__host__ __device__ inline size_t ws_doubles(int N, int M, int nbmax) { return ph_state_offset(N, M, nbmax) + ph_extra_doubles(M, nbmax); }

__host__ __device__ inline size_t ws_core_doubles(int N, int M, int nbmax) {
    QpDims d = make_dims(N, M, 0, nbmax);
    size_t n = 7 * d.nslot_max + 12 * (size_t)nbmax * d.oq + (size_t)(d.npb ? d.npb : 1) * d.oq + 3 * (size_t)nbmax * 3 * d.oq +
               2 * (size_t)d.nj * d.nk + (size_t)d.nj * d.ldb * d.ldb + 2 * (size_t)d.nj * (d.nk < 36 ? d.nk : 36) * (d.nk < 36 ? d.nk : 36) +
               (size_t)(d.nj > 1 ? d.nj - 1 : 1) * d.ldb * d.ldb + 4 +
               2 * (size_t)nbmax * M * 3 + 3 * (size_t)(M + 1) * 9 + M + 64 + (ws_int_count(N, M, nbmax) + 1) / 2 + 2 +
               /* polish: cand, V, S, counters, big factor */ polish_ws_doubles(d.nj, d.nk);
    return n;
}

__device__ inline QpWs carve(double* base, const QpDims& d, int nbmax) {
    QpDims dm = make_dims(d.N, d.M, 0, nbmax);
    QpWs w;
    double* p = base;
    w.s = p, p += dm.nslot_max;
    w.z = p, p += dm.nslot_max;
    w.s2 = p, p += dm.nslot_max;
    w.z2 = p, p += dm.nslot_max;
    w.rh = p, p += dm.nslot_max;
    w.cc = p, p += dm.nslot_max;
    w.ds = p, p += dm.nslot_max;
    w.cpacc = p, p += 12 * (size_t)nbmax * d.oq;
    w.pwgt = p, p += (size_t)(dm.npb ? dm.npb : 1) * d.oq;
    w.dx = p, p += (size_t)nbmax * 3 * d.oq;
    w.dxa = p, p += (size_t)nbmax * 3 * d.oq;
    w.cvec = p, p += (size_t)nbmax * 3 * d.oq;
    w.rbase = p, p += (size_t)dm.nj * dm.nk;
    w.rhs = p, p += (size_t)dm.nj * dm.nk;
    p += (4 - ((p - base) & 3)) & 3;  // 32-byte alignment of the blocks (vector loads of the tiled path)
    w.Td = p, p += (size_t)dm.nj * dm.ldb * dm.ldb;
    w.To = p, p += (size_t)(dm.nj > 1 ? dm.nj - 1 : 1) * dm.ldb * dm.ldb;
    w.Lf = p, p += 2 * (size_t)dm.nj * (dm.nk < 36 ? dm.nk : 36) * (dm.nk < 36 ? dm.nk : 36);  // wave path only (a short last batch)
    w.boxlo = p, p += (size_t)nbmax * d.M * 3;
    w.boxhi = p, p += (size_t)nbmax * d.M * 3;
    w.Lk = p, p += (size_t)(d.M + 1) * 9;
    w.Dk = p, p += (size_t)(d.M + 1) * 9;
    w.Ek = p, p += (size_t)(d.M + 1) * 9;
    w.segsc = p, p += d.M;
    int* ip = (int*)p;
    w.flist = ip, ip += (size_t)nbmax * d.M * d.N;
    w.fnear = ip, ip += (size_t)nbmax * d.M;
    w.fcnt = ip, ip += (size_t)nbmax * d.M;
    w.fbase = ip, ip += (size_t)nbmax * d.M;
    w.fperm = ip, ip += (size_t)nbmax * d.M;
    w.frank = ip, ip += (size_t)nbmax * d.M;
    w.tile_base = ip, ip += dm.ntile + 1;
    w.wi_of = ip, ip += (size_t)nbmax * d.oq;
    w.nrm = (float*)ip;
    w.polish = p + (ws_int_count(d.N, d.M, nbmax) + 1) / 2 + 2;
    w.prof = nullptr;
    return w;
}

// Q_base (rbp_planner.hpp:330-335) = int_0^1 B'''_i B'''_j
__constant__ double c_Qbase[36] = {720,  -1800, 1200,  0,     0,     -120, -1800, 4800,  -3600, 0,     600,   0,
                                   1200, -3600, 3600,  -1200, 0,     0,    0,     0,     -1200, 3600,  -3600, 1200,
                                   0,    600,   0,     -3600, 4800,  -1800, -120, 0,     0,     1200,  -1800, 720};
// Bernstein -> monomial, rows = control point, cols = descending powers (rbp_planner.hpp:338-343)
__constant__ double c_basis[36] = {-1, 5,   -10, 10, -5, 1, 5,  -20, 30, -20, 5, 0, -10, 30, -30, 10, 0, 0,
                                   10, -20, 10,  0,  0,  0, -5, 5,   0,  0,   0, 0, 1,   0,  0,   0,  0, 0};

__device__ inline size_t pair_index(int N, int qi, int qj) { return (size_t)qi * N - (size_t)qi * (qi + 1) / 2 + (qj - qi - 1); }

// ---- block reductions ------------------------------------------------------------------------------------------
// Inside a wave by DPP (quad permutes, half-row and row mirrors) and four readlanes -- six butterfly steps of ds_bpermute pairs are
// ~1000 cycles of dependent LDS round trips, and an interior-point iteration has ten of these reductions --, across the waves through
// `red`.  Two barriers: the one in front of the write would only protect the previous reduction's readers, which the barrier at ITS
// end has already let go.
template <int CTRL>
__device__ __forceinline__ double qp_dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double qp_red_op(double a, double b, int op) { return op == 0 ? a + b : (op == 1 ? fmax(a, b) : fmin(a, b)); }
__device__ __forceinline__ double qp_rl(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ inline double block_reduce(double v, int op /*0 sum,1 max,2 min*/, double* red) {
    v = qp_red_op(v, qp_dpp_f64<0xB1>(v), op);   // quad_perm [1,0,3,2]
    v = qp_red_op(v, qp_dpp_f64<0x4E>(v), op);   // quad_perm [2,3,0,1]
    v = qp_red_op(v, qp_dpp_f64<0x141>(v), op);  // row_half_mirror
    v = qp_red_op(v, qp_dpp_f64<0x140>(v), op);  // row_mirror: every lane of a row holds the row's result
    v = qp_red_op(qp_red_op(qp_rl(v, 0), qp_rl(v, 16), op), qp_red_op(qp_rl(v, 32), qp_rl(v, 48), op), op);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double r = red[0];
    for (int i = 1; i < QP_THREADS / 64; ++i) r = op == 0 ? r + red[i] : (op == 1 ? fmax(r, red[i]) : fmin(r, red[i]));
    __syncthreads();
    return r;
}

// ------------------------------------------------------------------------------------------------------------
// per-mission constants: L_j (continuity map), D_j = L_j'(2Q22)L_j + 2Q11, E_j = 2 Q12 L_{j+1}
// Aeq_base rows at knot j (rbp_planner.hpp:390-399):  dl^-i nn_i A_T.row(i) c_{j-1} = dr^-i nn_i A_0.row(i) c_j
// ------------------------------------------------------------------------------------------------------------
__device__ void mission_constants(const QpDims& d, const double* T, QpWs& w) {
    const int M = d.M;
    for (int m = threadIdx.x; m < M; m += QP_THREADS) w.segsc[m] = pow(T[m + 1] - T[m], -5.0);
    for (int j = threadIdx.x; j <= M; j += QP_THREADS) {
        double* L = w.Lk + 9 * j;
        for (int e = 0; e < 9; ++e) L[e] = 0;
        if (j >= 1 && j < M) {
            const double r = (T[j] - T[j - 1]) / (T[j + 1] - T[j]);  // dl / dr
            // c5 = u0 ; c4 = c5 - r (u1 - u0) ; c3 = 2 c4 - c5 + r^2 (u0 - 2 u1 + u2)   (rows: c3, c4, c5)
            L[0] = (1 + r) * (1 + r), L[1] = -2 * r * (1 + r), L[2] = r * r;
            L[3] = 1 + r, L[4] = -r, L[5] = 0;
            L[6] = 1, L[7] = 0, L[8] = 0;
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j <= M; j += QP_THREADS) {
        double* D = w.Dk + 9 * j;
        double* E = w.Ek + 9 * j;
        for (int e = 0; e < 9; ++e) D[e] = 0, E[e] = 0;
        if (j >= 1 && j < M) {
            const double sl = pow(T[j] - T[j - 1], -5.0), sr = pow(T[j + 1] - T[j], -5.0);  // build_Q_p :349-351
            const double* L = w.Lk + 9 * j;
            double QL[9];
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) {
                    double s = 0;
                    for (int c = 0; c < 3; ++c) s += c_Qbase[6 * (3 + a) + 3 + c] * sl * L[3 * c + b];
                    QL[3 * a + b] = s;
                }
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) {
                    double s = 0;
                    for (int c = 0; c < 3; ++c) s += L[3 * c + a] * QL[3 * c + b];
                    D[3 * a + b] = 2 * (s + c_Qbase[6 * a + b] * sr);
                }
            if (j + 1 < M) {
                const double* Ln = w.Lk + 9 * (j + 1);
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) {
                        double s = 0;
                        for (int c = 0; c < 3; ++c) s += c_Qbase[6 * a + 3 + c] * sr * Ln[3 * c + b];
                        E[3 * a + b] = 2 * s;  // rows u_j, cols u_{j+1}
                    }
            }
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------------------
// row sweeps.  Rows of a batch QP (G x <= h form, as in the oracle):
//   bound   (a,k,side,j6):  +x <= hi   /  -x <= -lo                       rbp_planner.hpp:626-635
//   frozen  (a,f,j6):       sg * n . x_a <= -rr + sg * n . dummy_f        :645-666   sg = +1 if a < f else -1
//   pair    (a<b,j6):       n . x_a - n . x_b <= -rr                      :668-679
// Control points j6 < 3 and j6 >= 6M-3 are pinned by the end-state equalities: their rows are constants and are
// only checked once (presolve).  Work item = one free control point of one batch agent (all its bound and frozen
// rows), then one (pair, control point).
// ------------------------------------------------------------------------------------------------------------
enum { PASS_INIT = 0, PASS_BUILD, PASS_AFF, PASS_STEP, PASS_PRESOLVE, PASS_CAND, PASS_VERIFY, PASS_CAND_GEO,
       PASS_UPBUILD /* step of iteration i (old state -> new state arrays) fused with BUILD of iteration i+1 */ };

struct PassIO {
    // inputs
    double mu0, s_floor, dreg, sigma_mu, alpha;
    // outputs (block-reduced by the caller)
    double sum0, sum1, sum2, vmax, vmin;
};

#define QP_POLISH_PART 1
#include "qp_polish.inc"
#undef QP_POLISH_PART

// (round 6) what a sweep looks up per control point before it can request anything -- its group, the group's row counts and normal offset,
// its tile's first slot -- copied ONCE per batch QP into LDS (by rank, i.e. already through fperm): in global memory these were two dependent
// trips per control point.  Only filled when the batch fits (wave path: nb <= 4); RowCtx::meta is null otherwise.
#define QP_META_G 160
struct SweepMeta {
    unsigned short fbase[QP_META_G];
    unsigned char grp[QP_META_G], near[QP_META_G], all[QP_META_G];
    int tile_base[16];
};
struct RowCtx {
    const SweepMeta* meta;  // LDS (see SweepMeta), or null: EVERY construction site sets it
    const PolishWs* pw;
    double* scal;  // this mission's diagnostic scalars (DevSession::scalars + mission * SC_N).  NOT a pointer to the DevSession:
                   // taking the kernel argument's address forces the whole struct into scratch memory
    int mission;
    int lds_avail;  // doubles of dynamic LDS behind the work-area pointer handed to the phases
    QpDims d;
    QpWs w;
    const double* ctrl;  // [N][3][oq] of this mission
    const float* normals;  // [npair][M][3]
    const double* radius;  // [N]
};

// 1/x for the row arithmetic of the sweeps: v_rcp_f64 plus two Newton steps (relative error ~1e-16 for normal x > 0).
// An IEEE division costs three times as many instructions (div_scale, div_fmas, div_fixup).
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
#if QP_RCP_NEWTON > 1
    r = fma(fma(-x, r, 1.0), r, r);
#endif
    return r;
}

// 1/sqrt(x): v_rsq_f64 plus two Newton steps (full double precision for normal x > 0); an IEEE sqrt followed by an IEEE
// division is ~6x the instructions and sits on the dependent chain of every Cholesky column
__device__ __forceinline__ double fast_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    y = fma(y, fma(-h * y, y, 0.5), y);
    y = fma(y, fma(-h * y, y, 0.5), y);
    return y;
}

// One row of G x <= h at slot r.  slack = h - g.x at the CURRENT control points (PASS_UPBUILD: at the trial point x + alpha dx),
// ga = g . dx_aff, gd = g . dx.  cw = 1 for rows that count in the sums, 0 for the second copy of a pair row.  Returns the
// Newton weight wgt, the right-hand-side scalar v and the multiplier zo where the pass defines them.  Step-length limits are
// tracked as io.vmax = max(-ds/s, -dz/z) (the caller takes the reciprocal), which needs no data-dependent division.
// The corrector term cc = ds_aff dz_aff and the step (ds, dz) are functions of (s, z, ga, gd): the STEP and UPBUILD sweeps
// recompute them instead of reading them back (three stored arrays fewer, see the note above QpWs).
// PRE: (s, z) of the row were fetched by the caller (the prefetch ring of the frozen-row stream, QP_ROW_PF)
template <int PASS, bool PRE = false>
__device__ __forceinline__ void row_op(double slack, double ga, double gd, size_t r, const QpWs& w, PassIO& io, double cw, double& wgt,
                                       double& v, double& zo, double s_in = 0.0, double z_in = 0.0) {
    if (PASS == PASS_INIT) {
        const double s = slack < io.s_floor ? io.s_floor : slack;
        w.s[r] = s;
        w.z[r] = io.mu0 / s;
    } else if (PASS == PASS_BUILD) {
        const double s = PRE ? s_in : QGC(w.s)[r], z = PRE ? z_in : QGC(w.z)[r];
        const double rg = s - slack;
        wgt = z * fast_rcp(s + io.dreg * z);  // = 1 / (s/z + dreg)
        v = -wgt * (rg - s);                  // predictor: rc / z = s
        zo = z;
        io.sum0 += cw * s * z;
        io.vmax = fmax(io.vmax, fabs(rg));
    } else if (PASS == PASS_AFF) {
        const double s = PRE ? s_in : QGC(w.s)[r], z = PRE ? z_in : QGC(w.z)[r];
        const double rg = s - slack;
        const double iz = fast_rcp(z), is = fast_rcp(s);
        wgt = z * fast_rcp(s + io.dreg * z);
        const double dza = wgt * (ga + rg - s);
        const double dsa = -s - s * dza * iz;  // (-s z - s dza) / z
        const double cc = dsa * dza;
        io.vmax = fmax(io.vmax, fmax(-dsa * is, -dza * iz));
        io.sum0 += cw * s * z, io.sum1 += cw * (s * dza + z * dsa), io.sum2 += cw * cc;
        // the corrector's right-hand side is affine in sigma*mu, which is only known after this sweep's reductions:
        //   v_corr = -wgt (rg - (s z + cc - sigma mu) / z) = v - sigma mu * wgt / z.   Both parts are accumulated here, so
        // the corrector needs no sweep of its own (out: v, and wgt := wgt / z)
        v = -wgt * (rg - s - cc * iz);
        wgt = wgt * iz;
    } else if (PASS == PASS_STEP) {
        const double s = PRE ? s_in : QGC(w.s)[r], z = PRE ? z_in : QGC(w.z)[r];
        const double rg = s - slack;
        const double iz = fast_rcp(z), is = fast_rcp(s);
        wgt = z * fast_rcp(s + io.dreg * z);
        const double dza = wgt * (ga + rg - s);
        const double cc = (-s - s * dza * iz) * dza;
        const double rcc = s * z + cc - io.sigma_mu;
        const double dz = wgt * (gd + rg - rcc * iz);
        const double ds = -(rcc + s * dz) * iz;
        io.vmax = fmax(io.vmax, fmax(-ds * is, -dz * iz));
    } else if (PASS == PASS_UPBUILD) {
        // old state (s, z) at the old point: slack_old = slack + alpha * gd
        const double s = PRE ? s_in : QGC(w.s)[r], z = PRE ? z_in : QGC(w.z)[r];
        const double rg = s - (slack + io.alpha * gd);
        const double iz = fast_rcp(z);
        const double w0 = z * fast_rcp(s + io.dreg * z);
        const double dza = w0 * (ga + rg - s);
        const double cc = (-s - s * dza * iz) * dza;
        const double rcc = s * z + cc - io.sigma_mu;
        const double dz = w0 * (gd + rg - rcc * iz);
        const double ds = -(rcc + s * dz) * iz;
        const double sn = s + io.alpha * ds, zn = z + io.alpha * dz;
        QG(w.s2)[r] = sn, QG(w.z2)[r] = zn;
        io.vmin = fmin(io.vmin, sn * zn);  // wide-neighbourhood test of the step just applied
        // ... and the next iteration's weights / residuals at the new point
        const double rgn = sn - slack;
        wgt = zn * fast_rcp(sn + io.dreg * zn);
        v = -wgt * (rgn - sn);
        zo = zn;
        io.sum0 += cw * sn * zn;
        io.vmax = fmax(io.vmax, fabs(rgn));
    } else if (PASS == PASS_PRESOLVE) {
        io.vmax = fmax(io.vmax, -slack);  // violation of a pinned (constant) row
    } else if (PASS == PASS_CAND) {
        const double s = PRE ? s_in : QGC(w.s)[r], z = PRE ? z_in : QGC(w.z)[r];
        wgt = (z > s || s < 1e-6) ? fmax(z / s, 1e-300) : 0.0;  // candidate for the active set; the value orders the warm start
        QG(w.cc)[r] = wgt;
        v = slack;
    } else if (PASS == PASS_CAND_GEO) {
        // candidates from the geometry alone (no interior-point iterate): rows within QP_WARM_TAU of their bound at the
        // current point.  Used from the second Gauss-Seidel pass on, where the current point is the previous pass's
        // optimum of this very batch and its active rows sit at slack ~ 0.
        wgt = slack < QP_WARM_TAU ? (slack < 1e-8 ? 1e3 : 1.0) : 0.0;
        QG(w.cc)[r] = wgt;
        v = slack;
    } else if (PASS == PASS_VERIFY) {
        const double sn = slack - gd;  // slack at x + dx
        QG(w.ds)[r] = sn;
        io.vmax = fmax(io.vmax, -sn);
        if (sn < -1e-11 && QGC(w.cc)[r] == 0.0) {  // violated row that is not a candidate yet
            QG(w.cc)[r] = 1.0;
            wgt = 1.0;
        }
    }
}

// One sweep over all rows of the batch QP.  Rows (G x <= h form, as in the oracle):
//   bound   (a,k,side,j6):  +x <= hi   /  -x <= -lo                       rbp_planner.hpp:626-635
//   pair    (a<b,j6):       n . x_a - n . x_b <= -rr                      :668-679
//   frozen  (a,f,j6):       sg * n . x_a <= -rr + sg * n . dummy_f        :645-666   sg = +1 if a < f else -1
// Control points j6 < 3 and j6 >= 6M-3 are pinned by the end-state equalities: their rows are constants and are only checked
// once (presolve).  Work item = one free control point of one batch agent with ALL its rows (see the note above QpWs).
// wi0 / wi_end / stride: the control points this thread takes (default: the whole batch QP on one workgroup; the chip-wide sweeps of the
// phase-split path hand every workgroup a range of tiles, see ph_sweep)
template <int PASS>
__device__ void row_pass(const RowCtx& c, PassIO& io, int wi0 = threadIdx.x, int wi_end = 1 << 30, int stride = QP_THREADS) {
    const QpDims& d = c.d;
    const QpWs& w = c.w;
    const int oq = d.oq, N = d.N, M = d.M, nb = d.nb;
    constexpr bool build = (PASS == PASS_BUILD || PASS == PASS_UPBUILD);
    constexpr bool aff = (PASS == PASS_AFF);  // S[0..2] / S[3..5] then hold the two parts of the corrector rhs (see row_op)
    constexpr bool accum = (build || aff);
    constexpr bool pinned_only = (PASS == PASS_PRESOLVE);
    constexpr bool cand = (PASS == PASS_CAND || PASS == PASS_CAND_GEO || PASS == PASS_VERIFY);
    constexpr bool need_da = (PASS == PASS_AFF || PASS == PASS_STEP || PASS == PASS_UPBUILD);
    constexpr bool need_dd = (PASS == PASS_STEP || PASS == PASS_VERIFY || PASS == PASS_UPBUILD);
    const int ncp = nb * oq;
    const int wi_stop = wi_end < ncp ? wi_end : ncp;
#ifndef QP_SNAKE
#define QP_SNAKE 0
#endif
    // (QP_SNAKE) tiles are ranked by falling row count and handed to the waves round robin: wave 0 gets the heaviest tile of every round.
    // Serpentine order -- odd rounds in reverse -- evens the waves' row counts out (sums differ in the last bits: another order of addition).
    constexpr bool snake_ok = QP_SNAKE && (PASS == PASS_BUILD || PASS == PASS_AFF || PASS == PASS_STEP || PASS == PASS_UPBUILD);
    const bool snake = snake_ok && wi0 == (int)threadIdx.x && stride == QP_THREADS;
    for (int rnd = 0;; ++rnd) {
        int wi;
        if (snake) {
            const int wv = (int)(threadIdx.x >> 6), nw = QP_THREADS / 64;
            if (rnd * QP_THREADS >= wi_stop) break;
            wi = rnd * QP_THREADS + (((rnd & 1) ? nw - 1 - wv : wv) << 6) + (int)(threadIdx.x & 63);
            if (wi >= wi_stop) continue;
        } else {
            wi = wi0 + rnd * stride;
            if (wi >= wi_stop) break;
        }
        constexpr bool pre_b = QP_ROW_BLK > 0 && QP_BLK_PRE && ((QP_BLK_MASK >> PASS) & 1) && (PASS == PASS_BUILD || PASS == PASS_AFF || PASS == PASS_STEP || PASS == PASS_UPBUILD);
        constexpr bool all_rows = (PASS == PASS_PRESOLVE || PASS == PASS_VERIFY || PASS == PASS_CAND_GEO || PASS == PASS_CAND);
        constexpr bool blk_pass = QP_ROW_BLK > 0 && ((QP_BLK_MASK >> PASS) & 1);  // passes that take the round-6 prologue (LDS look-ups, global loads, blocks)
        const __attribute__((address_space(3))) SweepMeta* mt = blk_pass ? (const __attribute__((address_space(3))) SweepMeta*)c.meta : nullptr;
        int grp, cnt_near, cnt_all, fb;
        size_t base;
        if (blk_pass && mt) {  // (uniform) the look-ups from LDS
            const int rk = wi / 6;
            grp = mt->grp[rk], cnt_near = mt->near[rk], cnt_all = mt->all[rk], fb = mt->fbase[rk];
            base = (size_t)mt->tile_base[wi >> 6] + (wi & 63);
        } else {
            grp = w.fperm[wi / 6];
            cnt_near = (all_rows && PASS != PASS_CAND) ? 0 : w.fnear[grp], cnt_all = all_rows ? w.fcnt[grp] : cnt_near, fb = w.fbase[grp];
            base = (size_t)w.tile_base[wi >> 6] + (wi & 63);
        }
        const int a = grp / M, seg = grp - a * M, i = wi % 6, j6 = 6 * seg + i, it = a * oq + j6;
        const bool pinned = (j6 < 3 || j6 >= oq - 3);
        if (pinned != pinned_only) continue;
        const int qa = d.first + a;
        double xa[3], da[3], dd[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (blk_pass) {
                xa[k] = QGC(c.ctrl)[((size_t)qa * 3 + k) * oq + j6];
                da[k] = need_da ? QGC(w.dxa)[((size_t)a * 3 + k) * oq + j6] : 0.0;
                dd[k] = need_dd ? QGC(w.dx)[((size_t)a * 3 + k) * oq + j6] : 0.0;
            } else {
                xa[k] = c.ctrl[((size_t)qa * 3 + k) * oq + j6];
                da[k] = need_da ? w.dxa[((size_t)a * 3 + k) * oq + j6] : 0.0;
                dd[k] = need_dd ? w.dx[((size_t)a * 3 + k) * oq + j6] : 0.0;
            }
        }
        double S[6] = {0, 0, 0, 0, 0, 0}, yv[3] = {0, 0, 0}, gz[3] = {0, 0, 0};
        const int cnt = all_rows ? cnt_all : cnt_near;
        const float* nr = w.nrm + (size_t)fb * 3;
        const size_t r0 = base + (size_t)d.ncol0 * 64;
        // (QP_ROW_BLK) the first block of the frozen-row stream is requested here, with everything else the control point starts from
        constexpr bool blk_path = QP_ROW_BLK > 0 && ((QP_BLK_MASK >> PASS) & 1);
        constexpr bool rd_sz_b = PASS == PASS_BUILD || PASS == PASS_AFF || PASS == PASS_STEP || PASS == PASS_UPBUILD || PASS == PASS_CAND;
        constexpr int BB = QP_ROW_BLK > 0 ? QP_ROW_BLK : 1;
        double cs[BB], cz[BB], ch[BB];
        float cn[BB][3];
        if (blk_path && cnt > 0) {
#pragma unroll
            for (int u = 0; u < BB; ++u) {
                const int ix = u < cnt ? u : cnt - 1;
                const size_t rx = r0 + (size_t)ix * 64;
                cs[u] = rd_sz_b ? QGC(w.s)[rx] : 0.0, cz[u] = rd_sz_b ? QGC(w.z)[rx] : 0.0, ch[u] = QGC(w.rh)[rx];
#pragma unroll
                for (int e = 0; e < 3; ++e) cn[u][e] = QGF(nr)[3 * ix + e];
            }
        }
        // ---- bound rows (idx 0..5)
        // (QP_ROW_BLK: their (s, z) are requested together up front -- in the update sweep every row's stores stand between it and the
        // next row's loads, which the compiler must not move across them)
        double bs[6] = {0, 0, 0, 0, 0, 0}, bz[6] = {0, 0, 0, 0, 0, 0};
        if (pre_b) {
#pragma unroll
            for (int e = 0; e < 6; ++e) bs[e] = QGC(w.s)[base + (size_t)e * 64], bz[e] = QGC(w.z)[base + (size_t)e * 64];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double hi = blk_pass ? QGC(w.boxhi)[((size_t)a * M + seg) * 3 + k] : w.boxhi[((size_t)a * M + seg) * 3 + k];
            const double lo = blk_pass ? QGC(w.boxlo)[((size_t)a * M + seg) * 3 + k] : w.boxlo[((size_t)a * M + seg) * 3 + k];
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const size_t r = base + (size_t)(2 * k + side) * 64;
                const double sg = side == 0 ? 1.0 : -1.0;
                const double slack = side == 0 ? hi - xa[k] : xa[k] - lo;
                double wgt = 0, v = 0, zo = 0;
                row_op<PASS, pre_b>(slack, sg * da[k], sg * dd[k], r, w, io, 1.0, wgt, v, zo, bs[2 * k + side], bz[2 * k + side]);
                if (cand && wgt != 0)
                    emit_cand(d, w, *c.pw, r, j6, a, -1, k == 0 ? sg : 0.0, k == 1 ? sg : 0.0, k == 2 ? sg : 0.0, slack,
                              (int)(((size_t)qa * 3 + k) * oq + j6), side == 0 ? hi : lo, wgt);
                if (accum) {
                    const int dg = k == 0 ? 0 : (k == 1 ? 3 : 5);  // diagonal slots of the packed 3x3
                    if (build) {
                        S[dg] += wgt;
                        gz[k] += sg * zo;
                        yv[k] += sg * v;
                    } else {
                        S[k] += sg * v, S[3 + k] += sg * wgt;
                    }
                }
            }
        }
        // ---- in-batch pair rows (idx 6 .. 6 + nb - 2): canonical orientation n . (x_hi - x_lo) >= rr, so that the copy in the
        // other agent's column sees bit-identical inputs
        auto pair_row = [&](auto pre_tag, int pb, double n0, double n1, double n2, const double (&xb)[3], const double (&fa)[3], const double (&fd)[3],
                            double rsum, double s_in, double z_in) {
            constexpr bool PRE_P = decltype(pre_tag)::value;
            const int b = pb < a ? pb : pb + 1;
            const bool a_lo = a < b;
            const size_t r = base + (size_t)(6 + pb) * 64;
            double gab = 0, gdb = 0;
            // e = x_hi - x_lo ; the G row is  n . x_lo - n . x_hi <= -rr
            const double e0 = a_lo ? xb[0] - xa[0] : xa[0] - xb[0], e1 = a_lo ? xb[1] - xa[1] : xa[1] - xb[1],
                         e2 = a_lo ? xb[2] - xa[2] : xa[2] - xb[2];
            const double slack = n0 * e0 + n1 * e1 + n2 * e2 - rsum;
            if (need_da) {
                const double f0 = fa[0], f1 = fa[1], f2 = fa[2];
                gab = a_lo ? n0 * (da[0] - f0) + n1 * (da[1] - f1) + n2 * (da[2] - f2) : n0 * (f0 - da[0]) + n1 * (f1 - da[1]) + n2 * (f2 - da[2]);
            }
            if (need_dd) {
                const double f0 = fd[0], f1 = fd[1], f2 = fd[2];
                gdb = a_lo ? n0 * (dd[0] - f0) + n1 * (dd[1] - f1) + n2 * (dd[2] - f2) : n0 * (f0 - dd[0]) + n1 * (f1 - dd[1]) + n2 * (f2 - dd[2]);
            }
            double wgt = 0, v = 0, zo = 0;
            row_op<PASS, PRE_P>(slack, gab, gdb, r, w, io, a_lo ? 1.0 : 0.0, wgt, v, zo, s_in, z_in);
            if (cand && wgt != 0 && a_lo) emit_cand(d, w, *c.pw, r, j6, a, b, n0, n1, n2, slack, -1, 0.0, wgt);
            if (accum) {
                const double sg = a_lo ? 1.0 : -1.0;  // coefficient of x_a in the row is sg * n
                if (build) {
                    if (a_lo) w.pwgt[(size_t)(a * nb - a * (a + 1) / 2 + (b - a - 1)) * oq + j6] = wgt;
                    S[0] += wgt * n0 * n0, S[1] += wgt * n0 * n1, S[2] += wgt * n0 * n2;
                    S[3] += wgt * n1 * n1, S[4] += wgt * n1 * n2, S[5] += wgt * n2 * n2;
                    const double zz = sg * zo, vv = sg * v;
                    gz[0] += zz * n0, gz[1] += zz * n1, gz[2] += zz * n2;
                    yv[0] += vv * n0, yv[1] += vv * n1, yv[2] += vv * n2;
                } else {
                    const double vv = sg * v, ww = sg * wgt;
                    S[0] += vv * n0, S[1] += vv * n1, S[2] += vv * n2;
                    S[3] += ww * n0, S[4] += ww * n1, S[5] += ww * n2;
                }
            }
        };
        if (pre_b && nb <= 4) {
            // (QP_ROW_BLK) up to three pair rows: everything they read -- normal, the other agent's point and directions, radii, (s, z) -- is
            // requested for all of them before the first is worked on (each cost two trips to memory, one after the other)
            float pn[3][3];
            double pxb[3][3], pfa[3][3], pfd[3][3], prs[3], pps[3], ppz[3];
#pragma unroll
            for (int pb = 0; pb < 3; ++pb) {
                const int pq = pb < nb - 1 ? pb : 0;  // (a row that does not exist reads row 0's operands and is not worked on; nb = 1: nothing is)
                const int b = nb > 1 ? (pq < a ? pq : pq + 1) : 0;
                const bool a_lo = a < b;
                const int qb = d.first + b;
                const size_t r = base + (size_t)(6 + pq) * 64;
                const size_t nvo = nb > 1 ? (pair_index(N, a_lo ? qa : qb, a_lo ? qb : qa) * M + seg) * 3 : 0;
#pragma unroll
                for (int e = 0; e < 3; ++e) pn[pb][e] = QGF(c.normals)[nvo + e];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    pxb[pb][k] = QGC(c.ctrl)[((size_t)qb * 3 + k) * oq + j6];
                    pfa[pb][k] = need_da ? QGC(w.dxa)[((size_t)b * 3 + k) * oq + j6] : 0.0;
                    pfd[pb][k] = need_dd ? QGC(w.dx)[((size_t)b * 3 + k) * oq + j6] : 0.0;
                }
                prs[pb] = QGC(c.radius)[qa] + QGC(c.radius)[qb];
                pps[pb] = QGC(w.s)[r], ppz[pb] = QGC(w.z)[r];
            }
#pragma unroll
            for (int pb = 0; pb < 3; ++pb)
                if (pb < nb - 1)
                    pair_row(std::true_type{}, pb, pn[pb][0], pn[pb][1], pn[pb][2], pxb[pb], pfa[pb], pfd[pb], prs[pb], pps[pb], ppz[pb]);
        } else {
            for (int pb = 0; pb < nb - 1; ++pb) {
                const int b = pb < a ? pb : pb + 1;
                const bool a_lo = a < b;
                const int qb = d.first + b;
                const float* nv = c.normals + (pair_index(N, a_lo ? qa : qb, a_lo ? qb : qa) * M + seg) * 3;
                const double n0 = nv[0], n1 = nv[1], n2 = nv[2];
                double xb[3], fa[3] = {0, 0, 0}, fd[3] = {0, 0, 0};
#pragma unroll
                for (int k = 0; k < 3; ++k) xb[k] = c.ctrl[((size_t)qb * 3 + k) * oq + j6];
                const double rsum = c.radius[qa] + c.radius[qb];
                if (need_da) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) fa[k] = w.dxa[((size_t)b * 3 + k) * oq + j6];
                }
                if (need_dd) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) fd[k] = w.dx[((size_t)b * 3 + k) * oq + j6];
                }
                pair_row(std::false_type{}, pb, n0, n1, n2, xb, fa, fd, rsum, 0.0, 0.0);
            }
        }
        // ---- frozen neighbours that survived the presolve (rows implied by the SFC box of this segment are dropped): a stream
        // over the SELL arrays; the signed normal is shared by the six rows of the group
        // PRESOLVE / VERIFY / CAND_GEO see every row; the passes of the interior-point loop the near ones (the near rows come first in a
        // group's list); CAND walks all of them to clear the far rows' candidate marks
        if constexpr (blk_path) {
        // Under load a trip to memory takes 3-5 us and a row's arithmetic 0.1: a thread that loads a row, works on it and stores it pays the
        // trip once per row (the loads of the next row cannot move above the stores of this one, and the compiler drains the memory counter
        // wherever a loop-carried load is used).  So the stream runs in BLOCKS: the (s, z), constant and normal of the NEXT QP_ROW_BLK rows
        // are requested before the current block is worked on, and first touched -- copied -- after it: one trip per block, overlapped with
        // the block's arithmetic.  Rows past the end re-read the last row (same cache lines) and are not used.  Same rows, same order, same
        // arithmetic per row as the plain loop.
        constexpr bool rd_sz = PASS == PASS_BUILD || PASS == PASS_AFF || PASS == PASS_STEP || PASS == PASS_UPBUILD || PASS == PASS_CAND;
        constexpr int B = BB;
        if (cnt > 0) {
            double ns[B], nz[B], nh[B];
            float nn[B][3];
            for (int i0 = 0; i0 < cnt; i0 += B) {
#pragma unroll
                for (int u = 0; u < B; ++u) {
                    const int ix = i0 + B + u < cnt ? i0 + B + u : cnt - 1;
                    const size_t rx = r0 + (size_t)ix * 64;
                    ns[u] = rd_sz ? QGC(w.s)[rx] : 0.0, nz[u] = rd_sz ? QGC(w.z)[rx] : 0.0, nh[u] = QGC(w.rh)[rx];
#pragma unroll
                    for (int e = 0; e < 3; ++e) nn[u][e] = QGF(nr)[3 * ix + e];
                }
#pragma unroll
                for (int u = 0; u < B; ++u) {
                    const int idx = i0 + u;
                    if (idx < cnt) {
                        const size_t r = r0 + (size_t)idx * 64;
                        if (PASS == PASS_CAND && idx >= cnt_near) {  // a far row: no (s, z); not a candidate unless a verification finds it violated
                            QG(w.cc)[r] = 0.0;
                        } else {
                            const double n0 = cn[u][0], n1 = cn[u][1], n2 = cn[u][2];
                            const double slack = ch[u] - (n0 * xa[0] + n1 * xa[1] + n2 * xa[2]);
                            double wgt = 0, v = 0, zo = 0;
                            row_op<PASS, rd_sz>(slack, n0 * da[0] + n1 * da[1] + n2 * da[2], n0 * dd[0] + n1 * dd[1] + n2 * dd[2], r, w, io, 1.0, wgt, v, zo, cs[u], cz[u]);
                            if (cand && wgt != 0) emit_cand(d, w, *c.pw, r, j6, a, -1, n0, n1, n2, slack, -1, 0.0, wgt);
                            if (accum) {
                                if (build) {
                                    S[0] += wgt * n0 * n0, S[1] += wgt * n0 * n1, S[2] += wgt * n0 * n2;
                                    S[3] += wgt * n1 * n1, S[4] += wgt * n1 * n2, S[5] += wgt * n2 * n2;
                                    gz[0] += zo * n0, gz[1] += zo * n1, gz[2] += zo * n2;
                                    yv[0] += v * n0, yv[1] += v * n1, yv[2] += v * n2;
                                } else {
                                    S[0] += v * n0, S[1] += v * n1, S[2] += v * n2;
                                    S[3] += wgt * n0, S[4] += wgt * n1, S[5] += wgt * n2;
                                }
                            }
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < B; ++u) {
                    cs[u] = ns[u], cz[u] = nz[u], ch[u] = nh[u];
#pragma unroll
                    for (int e = 0; e < 3; ++e) cn[u][e] = nn[u][e];
                }
            }
        }
        } else {
#if QP_ROW_PF > 0
        // Under load the stream is bound by the memory operations in flight per thread, and unrolling the (heavy) row arithmetic to get
        // more of them costs registers and instruction cache (unroll 4 measured 8 % slower than 2).  A prefetch ring decouples the two:
        // the loads of row idx + QP_ROW_PF -- (s, z), the row constant, the group normal: nine registers -- are issued while row idx is
        // worked on, the arithmetic is not replicated.
        constexpr bool rd_sz = PASS == PASS_BUILD || PASS == PASS_AFF || PASS == PASS_STEP || PASS == PASS_UPBUILD || PASS == PASS_CAND;
        double ps[QP_ROW_PF], pz[QP_ROW_PF], ph[QP_ROW_PF];
        float pn[QP_ROW_PF][3];
#pragma unroll
        for (int u = 0; u < QP_ROW_PF; ++u) {
            const int iu = u < cnt ? u : (cnt > 0 ? cnt - 1 : 0);
            const size_t ru = r0 + (size_t)iu * 64;
            ps[u] = (rd_sz && cnt > 0) ? w.s[ru] : 0.0, pz[u] = (rd_sz && cnt > 0) ? w.z[ru] : 0.0, ph[u] = cnt > 0 ? w.rh[ru] : 0.0;
#pragma unroll
            for (int e = 0; e < 3; ++e) pn[u][e] = cnt > 0 ? nr[3 * iu + e] : 0.0f;
        }
        for (int idx = 0; idx < cnt; ++idx) {
            const size_t r = r0 + (size_t)idx * 64;
            const bool far_row = PASS == PASS_CAND && idx >= cnt_near;
            const double n0 = pn[0][0], n1 = pn[0][1], n2 = pn[0][2], rhv = ph[0], s_in = ps[0], z_in = pz[0];
#pragma unroll
            for (int u = 0; u + 1 < QP_ROW_PF; ++u) {
                ps[u] = ps[u + 1], pz[u] = pz[u + 1], ph[u] = ph[u + 1];
#pragma unroll
                for (int e = 0; e < 3; ++e) pn[u][e] = pn[u + 1][e];
            }
            {
                const int ix = idx + QP_ROW_PF < cnt ? idx + QP_ROW_PF : cnt - 1;  // (past the end: the last row again, never used)
                const size_t rx = r0 + (size_t)ix * 64;
                if (rd_sz) ps[QP_ROW_PF - 1] = w.s[rx], pz[QP_ROW_PF - 1] = w.z[rx];
                ph[QP_ROW_PF - 1] = w.rh[rx];
#pragma unroll
                for (int e = 0; e < 3; ++e) pn[QP_ROW_PF - 1][e] = nr[3 * ix + e];
            }
            if (far_row) {  // (see the plain loop below)
                w.cc[r] = 0.0;
                continue;
            }
            const double slack = rhv - (n0 * xa[0] + n1 * xa[1] + n2 * xa[2]);
            double wgt = 0, v = 0, zo = 0;
            row_op<PASS, rd_sz>(slack, n0 * da[0] + n1 * da[1] + n2 * da[2], n0 * dd[0] + n1 * dd[1] + n2 * dd[2], r, w, io, 1.0, wgt, v, zo, s_in, z_in);
            if (cand && wgt != 0) emit_cand(d, w, *c.pw, r, j6, a, -1, n0, n1, n2, slack, -1, 0.0, wgt);
            if (accum) {
                if (build) {
                    S[0] += wgt * n0 * n0, S[1] += wgt * n0 * n1, S[2] += wgt * n0 * n2;
                    S[3] += wgt * n1 * n1, S[4] += wgt * n1 * n2, S[5] += wgt * n2 * n2;
                    gz[0] += zo * n0, gz[1] += zo * n1, gz[2] += zo * n2;
                    yv[0] += v * n0, yv[1] += v * n1, yv[2] += v * n2;
                } else {
                    S[0] += v * n0, S[1] += v * n1, S[2] += v * n2;
                    S[3] += wgt * n0, S[4] += wgt * n1, S[5] += wgt * n2;
                }
            }
        }
#else
#pragma unroll QP_ROW_UNROLL
        for (int idx = 0; idx < cnt; ++idx) {
            const size_t r = r0 + (size_t)idx * 64;
            if (PASS == PASS_CAND && idx >= cnt_near) {  // a far row: no (s, z); not a candidate unless a verification finds it violated
                w.cc[r] = 0.0;
                continue;
            }
            const double n0 = nr[3 * idx], n1 = nr[3 * idx + 1], n2 = nr[3 * idx + 2];
            const double slack = w.rh[r] - (n0 * xa[0] + n1 * xa[1] + n2 * xa[2]);
            double wgt = 0, v = 0, zo = 0;
            row_op<PASS>(slack, n0 * da[0] + n1 * da[1] + n2 * da[2], n0 * dd[0] + n1 * dd[1] + n2 * dd[2], r, w, io, 1.0, wgt, v, zo);
            if (cand && wgt != 0) emit_cand(d, w, *c.pw, r, j6, a, -1, n0, n1, n2, slack, -1, 0.0, wgt);
            if (accum) {
                if (build) {
                    S[0] += wgt * n0 * n0, S[1] += wgt * n0 * n1, S[2] += wgt * n0 * n2;
                    S[3] += wgt * n1 * n1, S[4] += wgt * n1 * n2, S[5] += wgt * n2 * n2;
                    gz[0] += zo * n0, gz[1] += zo * n1, gz[2] += zo * n2;
                    yv[0] += v * n0, yv[1] += v * n1, yv[2] += v * n2;
                } else {
                    S[0] += v * n0, S[1] += v * n1, S[2] += v * n2;
                    S[3] += wgt * n0, S[4] += wgt * n1, S[5] += wgt * n2;
                }
            }
        }
#endif
        }
        if (accum) {
            double* acc = w.cpacc + it;
#pragma unroll
            for (int e = 0; e < 6; ++e) acc[(size_t)e * ncp] = S[e];
            if (build) {
#pragma unroll
                for (int e = 0; e < 3; ++e) acc[(size_t)(6 + e) * ncp] = yv[e], acc[(size_t)(9 + e) * ncp] = gz[e];
            }
        }
    }
}

// A sweep of the interior-point loop: a stand-alone function (own register allocation, +3 %) that reads the row context from an LDS
// copy made once per batch QP (passing the 500-byte struct by value would travel through scratch memory).
template <int PASS>
__device__ __noinline__ PassIO sweep(const RowCtx* cl, PassIO io) {
    row_pass<PASS>(*cl, io);
    return io;
}
#define SWEEP(PASS) io = sweep<PASS>(&c_lds, io)

// ------------------------------------------------------------------------------------------------------------
// control-space <-> reduced-space maps
// ------------------------------------------------------------------------------------------------------------
// cvec[a][k][j6] (control space)  ->  out[(j-1)*nk + (a*3+k)*3 + e] = (F' cvec)
__device__ void apply_FT(const QpDims& d, const QpWs& w, const double* cvec, double* out, double scale) {
    const int oq = d.oq, nu = 3 * d.nb;
    for (int it = threadIdx.x; it < d.nj * nu; it += QP_THREADS) {
        const int j = it / nu + 1, u = it % nu;
        const double* xr = cvec + (size_t)u * oq + 6 * j;
        const double* xl = cvec + (size_t)u * oq + 6 * (j - 1) + 3;
        const double* L = w.Lk + 9 * j;
#pragma unroll
        for (int e = 0; e < 3; ++e)
            out[(size_t)(j - 1) * d.nk + u * 3 + e] = scale * (xr[e] + L[0 + e] * xl[0] + L[3 + e] * xl[1] + L[6 + e] * xl[2]);
    }
}
// dx[a][k][j6] = F du   (pinned control points get 0).  One work item per (knot, agent, dim): its three reduced values feed the six
// control points around the knot -- 420 items with a dozen independent loads each for a batch of four, two rounds of the workgroup,
// where a loop over the 2592 control-space entries was ten rounds of one load-wait-store each (~1 us per round under load).
__device__ void apply_F(const QpDims& d, const QpWs& w, const double* du, double* dx) {
    const int oq = d.oq, nu = 3 * d.nb;
    for (int it = threadIdx.x; it < d.nj * nu; it += QP_THREADS) {
        const int j = it / nu + 1, u = it % nu;
        const double* uu = du + (size_t)(j - 1) * d.nk + u * 3;
        const double* L = w.Lk + 9 * j;
        const double u0 = uu[0], u1 = uu[1], u2 = uu[2];
        double* o = dx + (size_t)u * oq + 6 * (j - 1) + 3;  // control points 6(j-1)+3 .. 6j+2
        o[0] = L[0] * u0 + L[1] * u1 + L[2] * u2;
        o[1] = L[3] * u0 + L[4] * u1 + L[5] * u2;
        o[2] = L[6] * u0 + L[7] * u1 + L[8] * u2;
        o[3] = u0, o[4] = u1, o[5] = u2;
    }
    for (int it = threadIdx.x; it < nu * 6; it += QP_THREADS) {  // the pinned ends
        const int u = it / 6, q = it % 6;
        dx[(size_t)u * oq + (q < 3 ? q : oq - 6 + q)] = 0.0;
    }
}

// rbase = -F'(2Qx + G'z) in one pass (grad_ctrl + apply_FT fused, as in rhs_from_acc below); also returns this thread's
// share of max|rbase| and max|2Qx + G'z| for the dual-residual test
__device__ void rbase_from_acc(const RowCtx& c, double& dmax, double& gmax) {
    const QpDims& d = c.d;
    const QpWs& w = c.w;
    const int oq = d.oq, nu = 3 * d.nb;
    dmax = gmax = 0;
    for (int it = threadIdx.x; it < d.nj * nu; it += QP_THREADS) {
        const int j = it / nu + 1, u = it % nu, a = u / 3, k = u % 3;
        const double* xb = c.ctrl + ((size_t)(d.first + a) * 3 + k) * oq;
        double g[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int j6 = 6 * (j - 1) + 3 + q, m = j6 / 6, i = j6 % 6;
            const double* xs = xb + 6 * m;
            double gv = 0;
#pragma unroll
            for (int jj = 0; jj < 6; ++jj) gv += c_Qbase[6 * i + jj] * xs[jj];
            gv *= 2 * w.segsc[m];
            gv += w.cpacc[(size_t)(9 + k) * (d.nb * oq) + (size_t)a * oq + j6];  // G'z of ALL rows of this control point (bounds, pairs, frozen)
            g[q] = -gv;
            gmax = fmax(gmax, fabs(gv));
        }
        const double* L = w.Lk + 9 * j;
        const size_t o0 = (size_t)(j - 1) * d.nk + u * 3;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const double r = g[3 + e] + L[0 + e] * g[0] + L[3 + e] * g[1] + L[6 + e] * g[2];
            w.rbase[o0 + e] = r;
            dmax = fmax(dmax, fabs(r));
        }
    }
}
// rhs = rbase + F'(G'v) in one pass: gtv_ctrl + apply_FT + the add fused (every control-space entry of G'v is used by
// exactly one reduced-space entry, so nothing is computed twice and two barriers and the cvec round trip disappear)
__device__ void rhs_from_acc(const RowCtx& c, bool corrector, double sigma_mu) {
    const QpDims& d = c.d;
    const QpWs& w = c.w;
    const int oq = d.oq, nu = 3 * d.nb;
    for (int it = threadIdx.x; it < d.nj * nu; it += QP_THREADS) {
        const int j = it / nu + 1, u = it % nu, a = u / 3, k = u % 3;
        double g[6];  // G'v at control points 6(j-1)+3 .. 6j+2 of (agent a, dim k)
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int j6 = 6 * (j - 1) + 3 + q;
            const double* ac = w.cpacc + (size_t)a * oq + j6;
            const size_t ncp = (size_t)d.nb * oq;
            const double gv = corrector ? ac[k * ncp] - sigma_mu * ac[(3 + k) * ncp] : ac[(6 + k) * ncp];
            g[q] = gv;
        }
        const double* L = w.Lk + 9 * j;
        const size_t o0 = (size_t)(j - 1) * d.nk + u * 3;
#pragma unroll
        for (int e = 0; e < 3; ++e) w.rhs[o0 + e] = w.rbase[o0 + e] + g[3 + e] + L[0 + e] * g[0] + L[3 + e] * g[1] + L[6 + e] * g[2];
    }
}

// ------------------------------------------------------------------------------------------------------------
// knot blocks:  T_j = blockdiag(D_j) + sum_p S_p (x) t_p t_p',   T_{j+1,j} = blockdiag(E_j')
// ------------------------------------------------------------------------------------------------------------
__device__ inline double sym3s(const double* S, size_t stride, int k, int l) {  // the same on a component-major array
    const int a = k < l ? k : l, b = k < l ? l : k;
    return S[(size_t)(a == 0 ? b : (a == 1 ? 2 + b : 5)) * stride];
}
__device__ inline double sym3(const double* S, int k, int l) {
    const int a = k < l ? k : l, b = k < l ? l : k;
    return S[a == 0 ? b : (a == 1 ? 2 + b : 5)];
}

// what the assembly of a knot block needs (a by-value subset of the row context: usable from the stand-alone factor function)
struct AsmArgs {
    const double *cpacc, *pwgt, *Lk, *Dk;
    const float* normals;
    double* Td;
    int N, M, nb, first, oq, ldb;
};

// one work item of the block assembly = (knot j, agent a, agent b, dim k, dim l): six 3x3-accumulator entries in, one 3x3 (e,f)
// tile of T_j out.  r = ((a * nb + b) * 3 + k) * 3 + l.  Ssum: LDS copy of the per-control-point weight sums [nb*oq][6] or nullptr.
// Off-diagonal (a != b) blocks carry -wgt n n' of the pair row (a, b): the weight was left in pwgt by the sweep (8 bytes per
// row, one independent load -- not a stored 3x3, not a chain of index loads)
// lower_only (wave path): lane r of a factor chain loads T[k][r] for every k but only ever uses k <= r (T is symmetric, ldl_rows works
// on the lower triangle), so the other half is neither computed nor written: half of the assembly work and of T's write traffic.  (The
// chain still LOADS whole rows: predicating its 36 loads and stores costs the dependent chain more than the bytes are worth, measured.)
__device__ __forceinline__ void assemble_load(const AsmArgs& A, const double* Ssum, int j, int a, int b, int k, int l, double (&Sv)[6]) {
    const int nb = A.nb, oq = A.oq;
#pragma unroll
    for (int p = 0; p < 6; ++p) {
        const int j6 = 6 * (j - 1) + 3 + p;
        double sv;
        if (a == b) {
            sv = Ssum ? sym3(Ssum + ((size_t)a * oq + j6) * 6, k, l) : sym3s(A.cpacc + (size_t)a * oq + j6, (size_t)nb * oq, k, l);
        } else {
            const int lo = a < b ? a : b, hi = a < b ? b : a, seg = j6 / 6;
            const float* nv = A.normals + (pair_index(A.N, A.first + lo, A.first + hi) * A.M + seg) * 3;
            sv = -A.pwgt[(size_t)(lo * nb - lo * (lo + 1) / 2 + (hi - lo - 1)) * oq + j6] * (double)nv[k] * (double)nv[l];
        }
        Sv[p] = sv;
    }
}
__device__ __forceinline__ void assemble_store(const AsmArgs& A, int j, int a, int b, int k, int l, const double (&Sv)[6], bool lower_only) {
    const int lb = A.ldb;
    const double* L = A.Lk + 9 * j;
    double* out = A.Td + (size_t)(j - 1) * lb * lb + (size_t)(a * 9 + k * 3) * lb + b * 9 + l * 3;
#pragma unroll
    for (int e = 0; e < 3; ++e)
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            double acc = Sv[0] * L[e] * L[f] + Sv[1] * L[3 + e] * L[3 + f] + Sv[2] * L[6 + e] * L[6 + f];
            if (e == f) acc += Sv[3 + e];
            if (a == b && k == l) acc += A.Dk[9 * j + 3 * e + f];
            if (!(lower_only && a == b && k == l && e > f)) out[(size_t)e * lb + f] = acc;
        }
}
__device__ __forceinline__ void assemble_item(const AsmArgs& A, const double* Ssum, int j, int r, bool lower_only = false) {
    const int nb = A.nb;
    const int a = r / (nb * 9), b = (r / 9) % nb, k = (r / 3) % 3, l = r % 3;
    if (lower_only && (a > b || (a == b && k > l))) return;
    double Sv[6];
    assemble_load(A, Ssum, j, a, b, k, l, Sv);
    assemble_store(A, j, a, b, k, l, Sv, lower_only);
}

// All knot blocks' needed halves at once (the 256-thread build, before the factorisation): the 3x3 tiles (A, B) = (3a + k, 3b + l) with
// A <= B are enumerated directly (no idle threads), and ASM_ILP tiles per thread are in flight at a time -- a tile is six dependent-free
// loads, a few dozen flops and up to nine stores, so one tile at a time is a chain of memory round trips.
#ifndef ASM_ILP
#define ASM_ILP 6  // (A/B on one box: 6 is 0.3 % ahead of 3)
#endif
__device__ void assemble_needed_halves(const AsmArgs& A, int nj) {
    const int n3 = 3 * A.nb, per = n3 * (n3 + 1) / 2, total = nj * per;
    for (int base = threadIdx.x; base < total; base += ASM_ILP * QP_THREADS) {
        double Sv[ASM_ILP][6];
        int jj[ASM_ILP], ta[ASM_ILP], tb[ASM_ILP];
#pragma unroll
        for (int u = 0; u < ASM_ILP; ++u) {
            const int it = base + u * QP_THREADS, itc = it < total ? it : total - 1;
            const int j = itc / per, t = itc - j * per;
            // row Ai of the upper-triangular enumeration: start(Ai) = Ai * n3 - Ai (Ai - 1) / 2 <= t
            int Ai = (int)(((2 * n3 + 1) - sqrtf((float)((2 * n3 + 1) * (2 * n3 + 1) - 8 * t))) * 0.5f);
            if (Ai * n3 - Ai * (Ai - 1) / 2 > t) Ai--;
            if ((Ai + 1) * n3 - (Ai + 1) * Ai / 2 <= t) Ai++;
            const int Bi = Ai + t - (Ai * n3 - Ai * (Ai - 1) / 2);
            jj[u] = j + 1, ta[u] = Ai, tb[u] = Bi;
            assemble_load(A, nullptr, j + 1, Ai / 3, Bi / 3, Ai % 3, Bi % 3, Sv[u]);
        }
#pragma unroll
        for (int u = 0; u < ASM_ILP; ++u)
            if (base + u * QP_THREADS < total) assemble_store(A, jj[u], ta[u] / 3, tb[u] / 3, ta[u] % 3, tb[u] % 3, Sv[u], true);
    }
}

// The same through the LDS, one wave per knot (r03): a knot's inputs -- the weight sums of its six control points for every batch agent,
// the pair weights, the pairs' normals of the two segments -- are fetched with coalesced loads into the wave's LDS scratch, the tiles are
// computed from there into a symmetric LDS image of the block, and the image leaves as whole rows (16 bytes per lane, 1 KB per store
// instruction).  Tile by tile from / to global memory a knot costs ~10^4 scattered 8-byte transactions, this way ~10^2 line requests --
// and under load the memory system is bound by transactions (see the row sweeps).  lds: QP_THREADS / 64 scratch areas of ASML_DOUBLES.
#define ASML_DOUBLES(nk, nb) ((nk) * KL_LD + (nb) * 36 + 3 * (nb) * ((nb) - 1) + 3 * (nb) * ((nb) - 1) + 8)
// One wave, a sequence of knots: next() returns the next knot (0-based) or -1, done(j) is called when T_j is on its way to global memory.
// scratch: ASML_DOUBLES(nk, nb) doubles of LDS of the wave's own.
// TO_GLOBAL = false: the block stays in the wave's scratch as a full symmetric image (rows KL_LD apart) for a reader in the same
// workgroup -- the 512-thread build's chains, see twisted_factor -- and next() must not return before that reader is done with the
// previous image.
template <bool TO_GLOBAL, class NextFn, class DoneFn>
__device__ __forceinline__ void assemble_knots_lds(const AsmArgs& A, double* scratch, NextFn next, DoneFn done) {
    const int nb = A.nb, oq = A.oq, nk = 9 * nb, npb = nb * (nb - 1) / 2, n3 = 3 * nb, per = n3 * (n3 + 1) / 2;
    const int lane = threadIdx.x & 63;
    const size_t ncp = (size_t)nb * oq;
    kl_lds* Timg = (kl_lds*)scratch;
    kl_lds* Sin = Timg + nk * KL_LD;  // [a][sym][p]
    kl_lds* Pw = Sin + nb * 36;      // [pair][p]
    kl_lds* Nr = Pw + 6 * npb;       // [pair][left / right segment][3]
    // this lane's tiles of a block's upper half (the same for every knot)
    int tA[2], tB[2], ntile = 0;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = lane + 64 * u, tc = t < per ? t : 0;
        int ai = (int)(((2 * n3 + 1) - sqrtf((float)((2 * n3 + 1) * (2 * n3 + 1) - 8 * tc))) * 0.5f);
        if (ai * n3 - ai * (ai - 1) / 2 > tc) ai--;
        if ((ai + 1) * n3 - (ai + 1) * ai / 2 <= tc) ai++;
        tA[u] = ai, tB[u] = ai + tc - (ai * n3 - ai * (ai - 1) / 2);
        if (t < per) ntile = u + 1;
    }
    const int row0 = (2 * lane) / nk, col0 = (2 * lane) % nk, drow = 128 / nk, dcol = 128 % nk;  // see the row copy below
    size_t nr_off = 0;  // this lane's entry of the pairs' normals (pair, left / right segment, component): the same for every knot
    if (lane < 6 * npb) {
        const int pw = lane / 6, sg = (lane / 3) % 2, c3 = lane % 3;
        int lo = 0, rest = pw;  // pair pw = (lo, hi) in the order lo * nb - lo (lo + 1) / 2 + (hi - lo - 1)
        while (rest >= nb - 1 - lo) rest -= nb - 1 - lo, lo++;
        const int hi = lo + 1 + rest;
        nr_off = (pair_index(A.N, A.first + lo, A.first + hi) * A.M + sg) * 3 + c3;
    }
    for (int j = next(); j >= 0; j = next()) {
        const int jn = j + 1, j60 = 6 * j + 3;  // first control point of the knot
        for (int idx = lane; idx < nb * 36; idx += 64) {
            const int a = idx / 36, e = (idx / 6) % 6, pp = idx % 6;
            Sin[idx] = A.cpacc[(size_t)e * ncp + (size_t)a * oq + j60 + pp];
        }
        for (int idx = lane; idx < 6 * npb; idx += 64) Pw[idx] = A.pwgt[(size_t)(idx / 6) * oq + j60 + idx % 6];
        if (lane < 6 * npb) Nr[lane] = (double)A.normals[nr_off + (size_t)j * 3];  // (6 npb <= 36: one entry per lane)
        double L[9], Dk[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) L[e] = A.Lk[9 * jn + e], Dk[e] = A.Dk[9 * jn + e];
        kl_sync();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u < ntile) {
                const int Ai = tA[u], Bi = tB[u], a = Ai / 3, k = Ai % 3, b = Bi / 3, l = Bi % 3;
                double Sv[6];
                if (a == b) {
                    const int kk = k < l ? k : l, ll = k < l ? l : k, sym = kk == 0 ? ll : (kk == 1 ? 2 + ll : 5);
#pragma unroll
                    for (int pp = 0; pp < 6; ++pp) Sv[pp] = Sin[(a * 6 + sym) * 6 + pp];
                } else {  // a < b
                    const int pw = a * nb - a * (a + 1) / 2 + (b - a - 1);
#pragma unroll
                    for (int pp = 0; pp < 6; ++pp) Sv[pp] = -Pw[pw * 6 + pp] * Nr[(pw * 2 + (pp >= 3)) * 3 + k] * Nr[(pw * 2 + (pp >= 3)) * 3 + l];
                }
#pragma unroll
                for (int e = 0; e < 3; ++e)
#pragma unroll
                    for (int f = 0; f < 3; ++f) {
                        double acc = Sv[0] * L[e] * L[f] + Sv[1] * L[3 + e] * L[3 + f] + Sv[2] * L[6 + e] * L[6 + f];
                        if (e == f) acc += Sv[3 + e];
                        if (Ai == Bi) acc += Dk[3 * e + f];
                        Timg[(3 * Ai + e) * KL_LD + 3 * Bi + f] = acc;
                        Timg[(3 * Bi + f) * KL_LD + 3 * Ai + e] = acc;
                    }
            }
        }
        kl_sync();
        if (TO_GLOBAL) {
            double* Tg = A.Td + (size_t)j * nk * nk;
            if ((nk & 1) == 0) {
                // (the factorisation uses the entries (r, k >= r) only: knot_ldl.  Row and column of a lane's pair advance by 128 elements
                // per round: carried along, not divided out -- nk is not a compile-time constant here, and two integer divisions per pair
                // were more instructions than everything else in this loop)
                int r = row0, k2 = col0;
                for (int idx = lane; idx < nk * nk / 2; idx += 64) {
                    if (QP_T_WRITE_FULL || k2 + 1 >= r) *(kl_d2*)(Tg + 2 * idx) = *(const kl_lds2*)(Timg + r * KL_LD + k2);
                    r += drow, k2 += dcol;
                    if (k2 >= nk) k2 -= nk, r++;
                }
            } else {
                for (int idx = lane; idx < nk * nk; idx += 64) Tg[idx] = Timg[(idx / nk) * KL_LD + idx % nk];
            }
        }
        done(j);
        kl_sync();
    }
}
// ---- just-in-time assembly by the chains' companion waves (256-thread build, round 6; see twisted_factor ROLE 2) --------------------------
// Two out-of-line steps per block, so that the companion's column loop keeps its registers and its stack frame to itself:
//   fasm_fetch   the block's inputs -- weight sums of its six control points per batch agent, pair weights, the pairs' normals -- travel from
//                global memory into the wave's LDS scratch by global_load_lds (4 bytes per lane and instruction, no registers, nobody waits):
//                issued right after the PREVIOUS block's tiles, they land while the companion follows that block's factorisation;
//   fasm_tiles   the 3 x 3 tiles from the LDS inputs into the symmetric LDS image the chain reads (the arithmetic of assemble_knots_lds).
// Scratch layout as in assemble_knots_lds (image | Sin | Pw | Nr), except that Nr holds the normals as the floats they are.
__device__ __forceinline__ AsmArgs uni(const AsmArgs& A);  // (arguments of a non-kernel function arrive in vector registers: see BlkArgs)
template <class T>
__device__ __forceinline__ T* uni(T* p);
__device__ __noinline__ void fasm_fetch(AsmArgs Av, double* scratch_v, int jv) {
    typedef __attribute__((address_space(1))) const void gvoid;
    typedef __attribute__((address_space(3))) void lvoid;
    typedef __attribute__((address_space(3))) char lchar;
    const AsmArgs A = uni(Av);
    double* scratch = uni(scratch_v);
    const int j = __builtin_amdgcn_readfirstlane(jv);
    const int nb = A.nb, oq = A.oq, nk = 9 * nb, npb = nb * (nb - 1) / 2, lane = threadIdx.x & 63, j60 = 6 * j + 3;
    const size_t ncp = (size_t)nb * oq;
    lchar* Sin = (lchar*)((kl_lds*)scratch + nk * KL_LD);
    lchar* Pw = Sin + (size_t)nb * 36 * 8;
    lchar* Nr = Pw + (size_t)6 * npb * 8;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the previous block's tiles have read their inputs)
    for (int c = 0; c * 64 < nb * 72; ++c) {  // Sin[a][sym][p], 8 bytes each, as dwords
        const int dw = c * 64 + lane, idx = dw >> 1, a = idx / 36, e = (idx / 6) % 6, pp = idx % 6;
        if (dw < nb * 72)
            __builtin_amdgcn_global_load_lds((gvoid*)((const unsigned*)(A.cpacc + (size_t)e * ncp + (size_t)a * oq + j60 + pp) + (dw & 1)), (lvoid*)(Sin + c * 256), 4, 0, 0);
    }
    for (int c = 0; c * 64 < npb * 12; ++c) {  // Pw[pair][p]
        const int dw = c * 64 + lane, idx = dw >> 1;
        if (dw < npb * 12)
            __builtin_amdgcn_global_load_lds((gvoid*)((const unsigned*)(A.pwgt + (size_t)(idx / 6) * oq + j60 + idx % 6) + (dw & 1)), (lvoid*)(Pw + c * 256), 4, 0, 0);
    }
    if (lane < 6 * npb) {  // Nr[pair][left / right segment][3] (floats)
        const int pw = lane / 6, sg = (lane / 3) % 2, c3 = lane % 3;
        int lo = 0, rest = pw;  // pair pw = (lo, hi) in the order lo * nb - lo (lo + 1) / 2 + (hi - lo - 1)
        while (rest >= nb - 1 - lo) rest -= nb - 1 - lo, lo++;
        const int hi = lo + 1 + rest;
        const size_t off = (pair_index(A.N, A.first + lo, A.first + hi) * A.M + sg + j) * 3 + c3;
        __builtin_amdgcn_global_load_lds((gvoid*)(A.normals + off), (lvoid*)Nr, 4, 0, 0);
    }
}
__device__ __noinline__ void fasm_tiles(AsmArgs Av, double* scratch_v, int jv) {
    const AsmArgs A = uni(Av);
    double* scratch = uni(scratch_v);
    const int j = __builtin_amdgcn_readfirstlane(jv);
    const int nb = A.nb, nk = 9 * nb, npb = nb * (nb - 1) / 2, n3 = 3 * nb, per = n3 * (n3 + 1) / 2, lane = threadIdx.x & 63, jn = j + 1;
    kl_lds* Timg = (kl_lds*)scratch;
    const kl_lds* Sin = Timg + nk * KL_LD;
    const kl_lds* Pw = Sin + nb * 36;
    const __attribute__((address_space(3))) float* Nr = (const __attribute__((address_space(3))) float*)(Pw + 6 * npb);
    double L[9], Dk[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) L[e] = A.Lk[9 * jn + e], Dk[e] = A.Dk[9 * jn + e];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the inputs have landed (fasm_fetch was issued a block ago: nothing to wait for)
    kl_sync();
    for (int u = 0; u * 64 < per; ++u) {
        const int t = lane + 64 * u;
        if (t < per) {
            int Ai = (int)(((2 * n3 + 1) - sqrtf((float)((2 * n3 + 1) * (2 * n3 + 1) - 8 * t))) * 0.5f);
            if (Ai * n3 - Ai * (Ai - 1) / 2 > t) Ai--;
            if ((Ai + 1) * n3 - (Ai + 1) * Ai / 2 <= t) Ai++;
            const int Bi = Ai + t - (Ai * n3 - Ai * (Ai - 1) / 2);
            const int a = Ai / 3, k = Ai % 3, b = Bi / 3, l = Bi % 3;
            double Sv[6];
            if (a == b) {
                const int kk = k < l ? k : l, ll = k < l ? l : k, sym = kk == 0 ? ll : (kk == 1 ? 2 + ll : 5);
#pragma unroll
                for (int pp = 0; pp < 6; ++pp) Sv[pp] = Sin[(a * 6 + sym) * 6 + pp];
            } else {  // a < b
                const int pw = a * nb - a * (a + 1) / 2 + (b - a - 1);
#pragma unroll
                for (int pp = 0; pp < 6; ++pp)
                    Sv[pp] = -Pw[pw * 6 + pp] * (double)Nr[(pw * 2 + (pp >= 3)) * 3 + k] * (double)Nr[(pw * 2 + (pp >= 3)) * 3 + l];
            }
#pragma unroll
            for (int e = 0; e < 3; ++e)
#pragma unroll
                for (int f = 0; f < 3; ++f) {
                    double acc = Sv[0] * L[e] * L[f] + Sv[1] * L[3 + e] * L[3 + f] + Sv[2] * L[6 + e] * L[6 + f];
                    if (e == f) acc += Sv[3 + e];
                    if (Ai == Bi) acc += Dk[3 * e + f];
                    Timg[(3 * Ai + e) * KL_LD + 3 * Bi + f] = acc;
                    Timg[(3 * Bi + f) * KL_LD + 3 * Ai + e] = acc;
                }
        }
    }
    kl_sync();
}
// all knots up front, one wave per knot in turn (256-thread build).  lds: QP_THREADS / 64 scratch areas.
__device__ void assemble_blocks_lds(const AsmArgs& A, int nj, double* lds) {
    const int wave = threadIdx.x >> 6, NW = QP_THREADS / 64;
    int j = wave - NW;
    assemble_knots_lds<true>(A, lds + (size_t)wave * ASML_DOUBLES(9 * A.nb, A.nb), [&] { j += NW; return j < nj ? j : -1; }, [](int) {});
}

__device__ inline AsmArgs asm_args(const RowCtx& c) {
    return AsmArgs{c.w.cpacc, c.w.pwgt, c.w.Lk, c.w.Dk, c.normals, c.w.Td, c.d.N, c.d.M, c.d.nb, c.d.first, c.d.oq, c.d.ldb};
}

// whole-workgroup assembly of all knot blocks (tiled path; the wave path assembles block by block BEHIND the factorisation
// chains, see twisted_factor)
__device__ void assemble_blocks(const RowCtx& c, double* lds) {
    const QpDims& d = c.d;
    const QpWs& w = c.w;
    const int nk = d.nk, oq = d.oq, nb = d.nb, lb = d.ldb;
    const AsmArgs A = asm_args(c);
    // stage 1: per (agent, control point) the 3x3 weight sum of ALL its rows (bounds, in-batch pairs, frozen neighbours) as the
    // sweep left it in cpacc, parked in LDS when it fits (every entry is read by nine (k, l) work items)
    const bool in_lds = nb * oq * 6 <= c.lds_avail;
    if (in_lds) {
        for (int it = threadIdx.x; it < nb * oq * 6; it += QP_THREADS) lds[it] = w.cpacc[(size_t)(it % 6) * (nb * oq) + it / 6];
        __syncthreads();
    }
    const int per_knot = nb * nb * 9;
    for (int it = threadIdx.x; it < d.nj * per_knot; it += QP_THREADS) assemble_item(A, in_lds ? lds : nullptr, it / per_knot + 1, it % per_knot);
    // the wave-register path and the LDS-resident tiled path build their coupling blocks from Ek directly
    if (d.nj > 1 && nk > 36 && 3 * lb * (lb + 2) > c.lds_avail) {
        const size_t noff = (size_t)(d.nj - 1) * lb * lb;
        for (size_t it = threadIdx.x; it < noff; it += QP_THREADS) {
            const int j = (int)(it / ((size_t)lb * lb)) + 1;  // couples knot j (cols) and j+1 (rows)
            const int rr = (int)(it % ((size_t)lb * lb)) / lb, cc = (int)(it % ((size_t)lb * lb)) % lb;
            double v = 0;
            if (rr / 3 == cc / 3 && rr < nk && cc < nk) v = w.Ek[9 * j + 3 * (cc % 3) + (rr % 3)];  // E_j' : rows u_{j+1}, cols u_j
            w.To[it] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// wave-register path (nk <= 36, i.e. batches of up to 4 agents): the whole block-tridiagonal Cholesky and the
// substitutions run in ONE wavefront with matrix rows held in VGPRs (lane r = row r, NK doubles per block) and
// v_readlane broadcasts instead of LDS traffic: every step of the dependent chains costs a few issue cycles
// instead of an LDS round trip, and no workgroup barrier is needed inside a knot.
// (Round 3: the cross-lane traffic of the factorisation goes through LDS broadcasts instead, see knot_lds.inc; v_readlane is
// left for the pivots.)  The substitutions stage the knots' M_j through LDS (prefetched by the otherwise idle waves).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double rl(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// TWISTED ("burn at both ends") factorisation: wave 0 eliminates blocks 0 .. mid-1 upwards, wave 1 eliminates blocks
// nj-1 .. mid+1 downwards, concurrently on two SIMDs; the middle block collects both Schur complements.  It halves the
// length of the dependent chain (factor and substitutions alike) at no extra arithmetic.
//   Lf[j] = M_j = L_j^-T (row r contiguous, zeros left of the diagonal), then 1 / d_j   (KF_STRIDE doubles per knot)
__device__ __forceinline__ int twist_mid(int nj) { return nj / 2; }

// Progress counters of the just-in-time block assembly (LDS ints behind the two chain areas): cnt[i] counts the helper waves that
// have finished the blocks of chain step i; a chain may load its i-th block when all QP_THREADS/64 - 2 of them have.
// Wave roles of the wave path's factorisation: 0, 1 = the two chains; 2, 3 = their companions (M = L^-T behind the chain, see
// wave_factor_follow); the 512-thread build has four more waves, which assemble the knot blocks behind the chains (one wave per block,
// through the LDS: assemble_knots_lds); the 256-thread build assembles all blocks up front (assemble_blocks_lds).
#define ASM_WAVE0 4                                                        // first assembling wave
#define ASM_HELPERS (QP_THREADS / 64 > ASM_WAVE0 ? QP_THREADS / 64 - ASM_WAVE0 : 0)   // assembling waves
// (round 6) 256-thread build: no waves to spare for the assembly -- but the two COMPANION waves (M = L^-T behind the chains) idle between
// the end of one block's factorisation and the start of the next (while the chain forms the coupling rows and the rank-36 update): each
// assembles its chain's NEXT block into an LDS image in that gap, just in time, and the chain reads it from there (the Timg path of the
// 512-thread build).  The up-front assembly phase -- 11 % of a mission's time under load, most of it spent waiting for its own loads and
// stores: profiles/r06_ab_qp_levers.txt -- and the round trip of every T_j through global memory disappear.
#ifndef QP_FOLLOW_ASM
#define QP_FOLLOW_ASM (ASM_HELPERS == 0)
#endif
#define ASMF_READY(cnt, h) ((kl_ldsi*)((cnt) + 76 + (h)))  // images the companion of chain h has finished (monotone over a factorisation)
__device__ __forceinline__ void wait_blocks(int* cnt, int i) {
    while (__hip_atomic_load(cnt + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < ASM_HELPERS) __builtin_amdgcn_s_sleep(2);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

typedef double d4 __attribute__((ext_vector_type(4)));

template <int NK>
__device__ __forceinline__ bool chol_rows(double (&a)[NK], double& dinv) {  // right-looking; a[c] = L[r][c] for lanes r >= c
    bool ok = true;                                                           // dinv: 1 / L[r][r] of this lane's row (tiled path)
    const int r = threadIdx.x & 63;
    dinv = 1.0;
#pragma unroll
    for (int c = 0; c < NK; ++c) {
        const double dcc = rl(a[c], c);
        if (!(dcc > 0)) ok = false;
        const double inv = fast_rsqrt(dcc);
        dinv = (r == c) ? inv : dinv;
        a[c] *= inv;
#pragma unroll
        for (int k = c + 1; k < NK; ++k) a[k] -= a[c] * rl(a[c], k);
    }
    return ok;
}

// ---- one chain of the twisted factorisation (round 3: knot_lds.inc, cross-lane traffic through LDS broadcasts) --------------------
// Per knot j of the chain (the chain's previous knot jp = j - dir):
//   S_j = T_j - X_j D_jp^-1 X_j'          X_j = T_{j,jp} L_jp^-T  (rows of block j), the rank-NK update on v_mfma_f64_16x16x4_f64
//   S_j = L_j D_j L_j'                    kl_ldl: column images in LDS
//   M_j = L_j^-T                          explicit (kl_row_times_LinvT): the substitutions are matrix-vector products with M_j, and
//   X_jn = Cpl M_j                        the coupling factor towards the next knot costs three multiply-adds per entry
// What a knot leaves in global memory for the substitutions is M_j (row r contiguous, entries k < r are zeros) and 1 / d_j: 5.5 KB of
// triangle + diagonal instead of the two full blocks (L_j, X_j: 20.7 KB) of the round-2 formulation; X never leaves the LDS.
#define KF_STRIDE(NK) ((NK) * (NK) + KL_I)  // doubles per knot in QpWs::Lf: M_j [NK][NK], then 1 / d_j

// coefficients of row rr of the coupling block between knot j and the next knot of the chain: T_{j+dir,j}[rr][3g + q], g = rr / 3
// (T_{j+1,j} = blockdiag(E_{knot j+1}'), T_{j-1,j} = T_{j,j-1}'; see assemble_blocks)
__device__ __forceinline__ void coupling_coef(const QpWs& w, int j, int dir, int rr, double& e0, double& e1, double& e2) {
    // (a GLOBAL load, not a flat one: a flat load also counts on the LDS counter, and the chain's next wait for its own LDS traffic would
    // wait for this trip to memory as well -- the very latency the early issue is meant to hide)
    const __attribute__((address_space(1))) double* E = QGC(w.Ek) + 9 * (dir > 0 ? j + 1 : j);
    e0 = dir > 0 ? E[rr % 3] : E[3 * (rr % 3)], e1 = dir > 0 ? E[3 + rr % 3] : E[3 * (rr % 3) + 1],
    e2 = dir > 0 ? E[6 + rr % 3] : E[3 * (rr % 3) + 2];
}

// the diagonal block of one knot: S = T_j (- U) into registers, factorised (column images in C, 1 / d in I); P / pbase: see kl_ldl.
// Timg != nullptr (512-thread build): T_j is read from the assembling wave's LDS image instead of global memory -- the block never leaves
// the CU, and the chain does not start every knot with a trip to memory --; when the values have arrived *consumed = consumed_value
// tells the assembling wave that it may overwrite the image.
template <int NK>
__device__ __forceinline__ bool knot_ldl(const QpWs& w, int j, bool minus_u, kl_lds* base, int r, bool act, int rr, kl_ldsi* P, int pbase,
                                         const kl_lds* Timg = nullptr, kl_ldsi* consumed = nullptr, int consumed_value = 0) {
    using A = KlArea<NK>;
    kl_lds *C = base + A::C, *I = base + A::I, *U = base + A::C;
    double a[NK];
    if (Timg) {
#pragma unroll
        for (int k = 0; k < NK; ++k) a[k] = Timg[k * KL_LD + rr];
        kl_sync();
        if (consumed) kl_publish(consumed, consumed_value);
    } else {
        const double* Tg = w.Td + (size_t)j * NK * NK;
        // T is symmetric: column access = row access; only k <= r was assembled and is used.  Not fetching the unused half (the lanes r < k
        // re-reading the diagonal entry of row k: -DQP_T_HALF_ROWS) saves 2.6 % of the kernel's HBM-side bytes and COSTS 2.7 % of its time (A/B on
        // one box, 85.6 k vs 87.9 k agent-trajectories/s): whole rows it is.
#pragma unroll
#ifdef QP_T_HALF_ROWS
        for (int k = 0; k < NK; ++k) a[k] = Tg[k * NK + (rr > k ? rr : k)];
#else
        for (int k = 0; k < NK; ++k) a[k] = Tg[k * NK + rr];
#endif
    }
    if (minus_u) {
#pragma unroll
        for (int k = 0; k < NK; ++k) a[k] -= U[rr * KL_LDU + k];
        kl_sync();
    }
    return kl_ldl<NK>(a, C, I, r, act, P, pbase);
}

// M = L^-T of the block just factorised: rows into MX (for the coupling factor) and, with 1 / d, into global memory (for the
// substitutions).  FOLLOW: run by the chain's companion wave concurrently with knot_ldl of the chain wave (kl_follow_LinvT).
template <int NK, bool FOLLOW>
__device__ __forceinline__ void knot_inverse(const QpWs& w, int j, kl_lds* base, int r, bool act, kl_ldsi* P, int pbase, int& seen, kl_ldsi* Mdone,
                                             int done_value) {
    using A = KlArea<NK>;
    kl_lds *C = base + A::C, *MX = base + A::MX, *I = base + A::I;
    double m[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) m[k] = (k == r) ? 1.0 : 0.0;
    if (FOLLOW)
        kl_follow_LinvT<NK>(m, C, I, P, pbase);
    else
        kl_row_times_LinvT<NK>(m, C, I);
    if (FOLLOW) kl_await_opaque(P, pbase + NK + 1);  // the LAST reciprocal pivot is written after the last image was announced
    const double dinv = I[act ? r : 0];  // (read before the chain is told to go on: its next block overwrites I)
    kl_store_rows<NK>(m, MX, r, act);
    if (FOLLOW) kl_publish(Mdone, done_value);
    __attribute__((address_space(1))) double* Mg = QG(w.Lf + (size_t)j * KF_STRIDE(NK));
    // (256-thread build: the next block's inputs, sent for a block ago by global_load_lds, are counted on this wave's memory counter; they have
    // long landed -- drain the counter HERE, so that the tiles' wait for them does not have to wait for the row stores below as well)
    if (FOLLOW && QP_FOLLOW_ASM) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (act) {  // row r of M_j and 1 / d_j -> QpWs::Lf (16 bytes per lane and store instruction: the store path of a CU is issue bound)
        __attribute__((address_space(1))) double* row = Mg + (size_t)r * NK;
        if ((NK & 1) == 0) {
#pragma unroll
            for (int k = 0; k < NK; k += 2)
                if (k + 1 >= r) *(__attribute__((address_space(1))) kl_d2*)(row + k) = kl_d2{m[k], m[k + 1]};  // (M_j is upper triangular; the staging does not fetch the rest)
        } else {
#pragma unroll
            for (int k = 0; k < NK; ++k)
                if (k >= r) row[k] = m[k];
        }
        Mg[NK * NK + r] = dinv;
    }
}

#if defined(QP_PROFILE) && !defined(QP_LHSTATS) && !defined(QP_SOLVE_TIMERS)  // chain-side timers of the left chain (100 MHz clock, like the phase timers): SC 25 = MFMA update, 26 = waiting for the
                   // block assembly, 27 = the knot itself (load, factorisation, waiting for M, coupling rows)
#define CHAIN_T0 long long ct_ = wall_clock64()
#define CHAIN_T(slot)                                                          \
    do {                                                                       \
        const long long t_ = wall_clock64();                                   \
        if (w.prof && dir > 0 && r == 0) w.prof[slot] += (double)(t_ - ct_);   \
        ct_ = t_;                                                              \
    } while (0)
#else
#define CHAIN_T0
#define CHAIN_T(slot)
#endif
// progress words of chain h (LDS ints behind the assembly counters): [2h] = images / pivots published by the chain wave (monotone over
// the whole factorisation: block i publishes i * (NK + 1) + 1 ...), [2h + 1] = blocks whose M rows the companion wave has put into MX
#define CHAIN_SYNC(cnt, h) ((kl_ldsi*)((cnt) + 64 + 2 * (h)))
// 512-thread build: [h] = number of blocks chain h has read out of the assembling waves' LDS images (a wave may overwrite its image of
// step i - 2 when this says i - 1)
#define ASM_CONSUMED(cnt, h) ((kl_ldsi*)((cnt) + 72 + (h)))

// one chain: blocks j0, j0+dir, ... (count of them).  The chain wave factorises; its companion wave (wave_factor_follow) computes
// M_j = L_j^-T a column or two behind and stores it.  On return the chain's LDS area holds the coupling factor X towards the middle
// block (MX) and the reciprocal pivots of its last block (I): wave_factor_mid reads both chains' areas.
template <int NK>
__device__ __forceinline__ bool wave_factor_chain(const QpDims& d, const QpWs& w, int j0, int count, int dir, double* ldsW, int* cnt, int h) {
    using A = KlArea<NK>;
    kl_lds* base = (kl_lds*)ldsW;
    kl_lds *MX = base + A::MX, *I = base + A::I, *U = base + A::C;
    kl_ldsi *P = CHAIN_SYNC(cnt, h), *Mdone = P + 1;
    const int r = threadIdx.x & 63;
    const bool act = r < NK;
    const int rr = act ? r : 0;
    bool ok = true;
    int seen = 0, seen_img = 0;
    for (int i = 0, j = j0; i < count; ++i, j += dir) {
        CHAIN_T0;
        if (i > 0) kl_syrk<NK>(MX, I, U, r, false);
        CHAIN_T(25);
        if (QP_FOLLOW_ASM)
            kl_await(ASMF_READY(cnt, h), i + 1, seen_img);  // the companion wave's image of this block
        else
            wait_blocks(cnt, i);
        CHAIN_T(26);
        double e0, e1, e2, x[NK];
        coupling_coef(w, j, dir, rr, e0, e1, e2);  // (issued here: three loads from the L2, ~1 us under load, needed after the factorisation)
        // (512-thread build: T_j from the LDS image of the assembling wave (side h, parity of the step), see twisted_factor)
        // (256-thread build: from the ONE image of the chain's companion wave, see twisted_factor ROLE 2)
        const kl_lds* Timg = ASM_HELPERS > 0 ? (const kl_lds*)(ldsW - (size_t)h * A::SIZE + 2 * A::SIZE + 128 + (size_t)(h + 2 * (i & 1)) * ASML_DOUBLES(NK, (NK / 9)))
                             : QP_FOLLOW_ASM ? (const kl_lds*)(ldsW - (size_t)h * A::SIZE + 2 * A::SIZE + 128 + (size_t)h * ASML_DOUBLES(NK, (NK / 9)))
                                             : nullptr;
        if (!knot_ldl<NK>(w, j, i > 0, base, r, act, rr, P, i * (NK + 1), Timg, ASM_CONSUMED(cnt, h), i + 1)) ok = false;
        kl_await(Mdone, i + 1, seen);
        kl_coupling_rows<NK>(x, MX, r, act, e0, e1, e2);
        CHAIN_T(27);
    }
    return ok;
}

// the middle block collects both Schur complements (the chains' areas: ldsL = left chain = this wave's own area, ldsR = right chain)
template <int NK>
__device__ __forceinline__ bool wave_factor_mid(const QpDims& d, const QpWs& w, double* ldsL, double* ldsR, int* cnt, int cnt_idx) {
    using A = KlArea<NK>;
    kl_lds *bl = (kl_lds*)ldsL, *br = (kl_lds*)ldsR;
    const int r = threadIdx.x & 63, mid = twist_mid(d.nj);
    const bool act = r < NK;
    const int rr = act ? r : 0;
    int nsy = 0;
    if (mid > 0) kl_syrk<NK>(bl + A::MX, bl + A::I, bl + A::C, r, false), nsy++;
    if (mid + 1 < d.nj) kl_syrk<NK>(br + A::MX, br + A::I, bl + A::C, r, nsy > 0), nsy++;
    if (QP_FOLLOW_ASM) {
        int seen_img = 0;
        kl_await(ASMF_READY(cnt, 0), mid + 1, seen_img);  // the left chain's companion makes the middle block after the chain's mid blocks
    } else {
        wait_blocks(cnt, cnt_idx);
    }
    // progress words of the middle block.  512-thread build: its M is computed by the left chain's companion wave, like every other
    // block's (twisted_factor, ROLE 2) -- on the chain wave it is ~10 k cycles more at the end of every factorisation: one mission
    // 122.4 -> 121.1 ms.  With two workgroups per CU the polling companion costs the neighbour more than it saves (-1.3 % at 2000
    // resident, A/B on one box): the 256-thread build keeps M on the chain wave.
    kl_ldsi* scratch = CHAIN_SYNC(cnt, 2);
    const kl_lds* Timg = ASM_HELPERS > 0 ? (const kl_lds*)(ldsL + 2 * A::SIZE + 128 + (size_t)(2 * (cnt_idx & 1)) * ASML_DOUBLES(NK, (NK / 9)))
                         : QP_FOLLOW_ASM ? (const kl_lds*)(ldsL + 2 * A::SIZE + 128) : nullptr;
    const bool ok = knot_ldl<NK>(w, mid, nsy > 0, bl, r, act, rr, scratch, 0, Timg);
    if (!QP_MID_FOLLOW) {
        int seen = 0;
        knot_inverse<NK, false>(w, mid, bl, r, act, scratch, 0, seen, scratch, 0);
    }
    return ok;
}

// whole twisted factorisation; every thread of the workgroup calls it.  flag: LDS int.
// The knot blocks T_j are ASSEMBLED HERE, by waves that do not run a chain, in the order the chains consume them (step i: blocks i
// and nj-1-i; last the middle one), each step announced through an LDS counter (wait_blocks): the assembly -- 8-10 % of an
// interior-point iteration when it was a phase of its own -- disappears behind the dependent chains.
// ROLE: 0 = compiled for the two chain waves, 2 = for their companions (waves 2, 3), 1 = for the assembling waves (4..) of the
// 512-thread build: three __noinline__ functions with the same workgroup barriers, see solve_staged.
template <int NK, int ROLE>
__device__ __forceinline__ bool twisted_factor(const QpDims& d, const QpWs& w, int* flag, double* lds, const AsmArgs* asmb) {
    const int wave = (ROLE == 0 || ROLE == 2) ? (threadIdx.x >> 6) & 1 : 4, mid = twist_mid(d.nj);  // roles 0 / 2: which chain
    const int nl = mid, nr = d.nj - 1 - mid, SF = nl > nr ? nl : nr;
    constexpr int AREA = KlArea<NK>::SIZE;    // one chain wave's LDS area (knot_lds.inc)
    int* cnt = (int*)(lds + 2 * AREA);  // [SF + 1] assembly counters, then (at +64) the chains' progress words
    static_assert((QP_MAX_M - 1) / 2 + 1 <= 64, "cnt[0..SF] must end below CHAIN_SYNC (cnt + 64): sessions with M > QP_MAX_M are refused");
    if (threadIdx.x == 0) *flag = 0;
    __syncthreads();
    bool ok = true;
    // rows >= NK of the X images are never written by a row's own lane: clear them once (they only feed unused tile entries)
    for (int i = threadIdx.x; i < 2 * AREA + 64; i += QP_THREADS) lds[i] = 0.0;
    __syncthreads();
    // The chain waves issue dependent instructions; whenever the SIMD's arbiter makes one of them queue behind the ready instructions
    // of other waves (the helpers, the co-resident workgroup's sweeps) the chain stretches, while giving it the first slot costs the
    // others next to nothing: raise the priority for the chain (+2.5 % at 2000 missions).
    if (ROLE == 0) {
        __builtin_amdgcn_s_setprio(QP_CHAIN_PRIO);
        if (wave == 0 && mid > 0) ok = wave_factor_chain<NK>(d, w, 0, mid, +1, lds, cnt, 0);
        if (wave == 1 && nr > 0) ok = wave_factor_chain<NK>(d, w, d.nj - 1, nr, -1, lds + AREA, cnt, 1);
        if (wave == 1) __builtin_amdgcn_s_setprio(0);
    }
    if (ROLE == 1) {
        // assembling waves (512-thread build): through the LDS like the 256-thread build's up-front assembly (assemble_knots_lds), one
        // wave per block.  Waves 4, 5 take the left / right chain's block of the even steps, waves 6, 7 those of the odd steps (two
        // chain steps of time per block); a wave announces its block by adding ASM_HELPERS / 2 to the step's counter, so that
        // wait_blocks sees ASM_HELPERS when both blocks of a step are out.  The middle block (step SF) is wave 4's / 6's.
        const AsmArgs A = *asmb;
        const int hw = (threadIdx.x >> 6) - ASM_WAVE0, side = hw & 1, par = hw >> 1;  // par: parity of the steps this wave serves
        double* scratch = lds + 2 * AREA + 128 + (size_t)hw * ASML_DOUBLES(NK, (NK / 9));
        int i = par - 2;
        kl_ldsi* consumed = ASM_CONSUMED(cnt, side);
        int seen_c = 0;
        assemble_knots_lds<false>(
            A, scratch,
            [&] {
                for (i += 2; i <= SF; i += 2) {
                    const bool work = i < SF ? (side ? i < nr : i < nl) : side == 0;
                    if (work) {
                        if (i >= 2) kl_await(consumed, i - 1, seen_c);  // the chain has read this wave's image of step i - 2
                        return i < SF ? (side ? d.nj - 1 - i : i) : mid;
                    }
                    // nothing to assemble for this wave at step i: announce it all the same
                    if ((threadIdx.x & 63) == 0)
                        __hip_atomic_fetch_add(cnt + i, i < SF ? ASM_HELPERS / 2 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                return -1;
            },
            [&](int) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if ((threadIdx.x & 63) == 0)
                    __hip_atomic_fetch_add(cnt + i, i < SF ? ASM_HELPERS / 2 : ASM_HELPERS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            });
    }
    if (ROLE == 2) {
        // companion of chain `wave`: M_j behind the chain's factorisation of block j
        const int count = wave == 0 ? nl : nr, j0 = wave == 0 ? 0 : d.nj - 1, dir = wave == 0 ? +1 : -1;
        kl_lds* base = (kl_lds*)(lds + wave * AREA);
        kl_ldsi *P = CHAIN_SYNC(cnt, wave), *Mdone = P + 1;
        const int r = threadIdx.x & 63;
        int seen = 0;
        if (QP_FOLLOW_ASM) {
            // ... and the chain's blocks, assembled just in time: block n + 1 right after the companion work of block n, while the chain
            // forms the coupling rows and the rank-NK update; the left chain's companion makes the MIDDLE block last.  One image per chain:
            // the chain read block n out of it at the start of its factorisation (ASM_CONSUMED), long before block n + 1 is written.
            const AsmArgs A = *asmb;
            double* scratch = lds + 2 * AREA + 128 + (size_t)wave * ASML_DOUBLES(NK, (NK / 9));
            kl_ldsi *ready = ASMF_READY(cnt, wave), *consumed = ASM_CONSUMED(cnt, wave);
            const int total = count + (wave == 0 ? 1 : 0);
            int seen_c = 0;
            auto blk = [&](int n) { return n < count ? j0 + n * dir : mid; };
            auto make = [&](int n) {  // block n's inputs are in the scratch (fetched a block ago): its tiles, then the fetch of block n + 1
                if (n > 0) kl_await(consumed, n, seen_c);
                fasm_tiles(A, scratch, blk(n));
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                kl_publish(ready, n + 1);
                if (n + 1 < total) fasm_fetch(A, scratch, blk(n + 1));
            };
            if (total > 0) {
                fasm_fetch(A, scratch, blk(0));
                make(0);
            }
            for (int i = 0; i < count; ++i) {
                knot_inverse<NK, true>(w, j0 + i * dir, base, r, r < NK, P, i * (NK + 1), seen, Mdone, i + 1);
                if (i + 1 < total) make(i + 1);
            }
        } else {
            for (int i = 0; i < count; ++i) knot_inverse<NK, true>(w, j0 + i * dir, base, r, r < NK, P, i * (NK + 1), seen, Mdone, i + 1);
        }
    }
    if (!ok && (threadIdx.x & 63) == 0) atomicExch(flag, 1);
    __threadfence_block();
    __syncthreads();
    if (*flag) return false;
    if (ROLE == 0 && wave == 0) {
        if (!wave_factor_mid<NK>(d, w, lds, lds + AREA, cnt, SF) && threadIdx.x == 0) *flag = 1;
        __builtin_amdgcn_s_setprio(0);
    }
    if (QP_MID_FOLLOW && ROLE == 2 && wave == 0) {  // M of the middle block, behind wave_factor_mid's factorisation
        const int r = threadIdx.x & 63;
        int seen = 0;
        kl_ldsi* P = CHAIN_SYNC(cnt, 2);
        knot_inverse<NK, true>(w, mid, (kl_lds*)lds, r, r < NK, P, 0, seen, P + 1, 1);
    }
    __threadfence_block();
    __syncthreads();
    return *flag == 0;
}

// Substitutions T du = rhs for the twisted factorisation (round 3: matrix-vector products with the explicit M_j = L_j^-T instead of
// 36-step dependent triangular solves).  With X_j = T_{j,jp} M_jp (jp = the chain's previous knot, never stored):
//   forward   r'  = rhs_j - T_{j,jp} v_jp          (3x3-block sparse)         y = M_j' r'      z_j = D_j^-1 y      v_j = M_j z_j
//   backward  w   = T_{jn,j}' x_jn                 (jn = next knot towards the middle)
//             x_j = M_j (z_j - D_j^-1 M_j' w)
// i.e. two dense products with the SAME staged matrix per chain step (one by columns, one by rows), vectors broadcast from LDS.
// The knots' M_j are STAGED THROUGH LDS: waves 2.. prefetch the blocks of step s + QP_STAGE_BUFS - 1 (coalesced global reads) while
// waves 0 / 1 run step s of the left / right chain.  rhs -> z -> x lives in LDS throughout (vec).
// ROLE: 0 = compiled for the two chain waves, 1 = for the staging waves (2..): two __noinline__ functions (solve_entry_*), so that the
// staging waves -- a dozen registers -- have no prologue saving callee-saved VGPRs.
// The two products of a chain step use M_j's triangle with ALL 64 lanes: a lane owns a PIECE -- KsPieces::CH consecutive terms of one row
// (or column) -- instead of a whole row, of which only NK <= 36 exist and whose lower half is zeros: 15 loads and multiply-adds per
// lane instead of 36, then the up to three pieces of a row meet through 64 doubles of LDS.  Chunk t of the product by rows covers the
// columns NK - CH (t + 1) .. NK - 1 - CH t (anchored at the right edge: row r needs the chunks with NK - 1 - CH t >= r), chunk t of the
// product by columns the rows CH t .. CH t + CH - 1 (anchored at the top: column c needs the chunks with CH t <= c).  Terms that fall
// left of / below the diagonal multiply the stage buffers' zeros, terms outside the block multiply the zero padding of the VECTOR (KS_VPAD
// slots either side; whatever finite number sits at the matrix address): no masks, no branches.
#define KS_VPAD 16                       // zero slots either side of a chain vector
#define KS_VLEN (KS_VPAD + KL_I + KS_VPAD)  // doubles per vector (the slot NK + 12 takes the writes of the lanes without a row)
template <int NK>
struct KsPieces {
    static constexpr int CH = (NK * 15 + 35) / 36;  // 36 -> 15, 27 -> 12, 18 -> 8, 9 -> 4
    static constexpr int N0 = NK, N1 = NK - CH > 0 ? NK - CH : 0, N2 = NK - 2 * CH > 0 ? NK - 2 * CH : 0;
    static_assert(NK - 3 * CH <= 0 && N0 + N1 + N2 <= 64, "three chunks, one wavefront");
    static_assert(3 * CH - NK <= KS_VPAD && 3 * CH - 1 - (NK - 1) <= KS_VPAD, "vector padding");
    __device__ static __forceinline__ void of_lane(int lane, int& idx, int& t) {
        t = lane < N0 ? 0 : (lane < N0 + N1 ? 1 : 2);
        idx = lane < N0 ? lane : (lane < N0 + N1 ? lane - N0 : (lane < N0 + N1 + N2 ? lane - N0 - N1 : 0));
    }
};
// sum_k Mst[k][c] * b[k] for column c = rr (lanes rr < NK; b: vector base, i.e. entry 0); psum: 64 doubles of scratch
template <int NK>
__device__ __forceinline__ double kl_matvec_cols(const kl_lds* Mst, const kl_lds* b, kl_lds* psum, int lane) {
    using P = KsPieces<NK>;
    int idx, t;
    P::of_lane(lane, idx, t);
    const int k0 = P::CH * t, c = k0 + idx;
    const kl_lds* mp = Mst + k0 * KL_LD + c;
    const kl_lds* bp = b + k0;
    double mv[P::CH], bv[P::CH];
#pragma unroll
    for (int j = 0; j < P::CH; ++j) mv[j] = mp[j * KL_LD], bv[j] = bp[j];
    double s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
    for (int j = 0; j < P::CH; ++j) {
        if (j % 3 == 0) s0 += mv[j] * bv[j];
        if (j % 3 == 1) s1 += mv[j] * bv[j];
        if (j % 3 == 2) s2 += mv[j] * bv[j];
    }
    psum[lane] = (s0 + s1) + s2;
    kl_sync();
    const int cc = lane < NK ? lane : 0;
    double r = psum[cc];
    if (P::N1 > 0) r += cc >= P::CH ? psum[P::N0 + cc - P::CH] : 0.0;
    if (P::N2 > 0) r += cc >= 2 * P::CH ? psum[P::N0 + P::N1 + cc - 2 * P::CH] : 0.0;
    return r;
}
// sum_k Mst[r][k] * b[k] for row r = rr
template <int NK>
__device__ __forceinline__ double kl_matvec_rows(const kl_lds* Mst, const kl_lds* b, kl_lds* psum, int lane) {
    using P = KsPieces<NK>;
    int idx, t;
    P::of_lane(lane, idx, t);
    const int k0 = NK - P::CH * (t + 1);  // may be negative: the vector is padded, the matrix address holds some finite number
    const kl_lds* mp = Mst + idx * KL_LD + k0;
    const kl_lds* bp = b + k0;
    double mv[P::CH], bv[P::CH];
#pragma unroll
    for (int j = 0; j < P::CH; ++j) mv[j] = mp[j], bv[j] = bp[j];
    double s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
    for (int j = 0; j < P::CH; ++j) {
        if (j % 3 == 0) s0 += mv[j] * bv[j];
        if (j % 3 == 1) s1 += mv[j] * bv[j];
        if (j % 3 == 2) s2 += mv[j] * bv[j];
    }
    psum[lane] = (s0 + s1) + s2;
    kl_sync();
    const int rr = lane < NK ? lane : 0;
    double r = psum[rr];
    if (P::N1 > 0) r += rr < P::N1 ? psum[P::N0 + rr] : 0.0;
    if (P::N2 > 0) r += rr < P::N2 ? psum[P::N0 + P::N1 + rr] : 0.0;
    return r;
}

template <int NK>
struct KsLayout {  // doubles
    static constexpr int SM = NK * KL_LD, SCH = SM + KL_I, STG = 2 * SCH;  // a stage: left chain (M, 1/d), right chain (M, 1/d)
    static constexpr int LEAD = 16;  // zeros in front of the first stage buffer (a row piece may start a few entries before its block)
    // per chain: A (r' / w), Z (z / u), V (v / x of the block just done), each with zero padding either side, and the pieces' scratch
    static constexpr int VECS = 6 * KS_VLEN + 2 * 64;
};

// (round 6) the solution's way back to control space (apply_F) can start from the LDS vector the substitutions end with, instead of from a copy
// in global memory written by one loop and read back by the next: dx_out != null fuses it into the epilogue (wave path only)
struct SolveOut {
    double* dx_out;    // [nb][3][oq] direction in control space, or null: the reduced solution goes to rhs as before
    const double* Lk;  // QpWs::Lk
    int oq;
    // ... and the right-hand side (rhs_from_acc) can be formed straight into that LDS vector: rhs_mode 1 = predictor, 2 = corrector, 0 = read rhs
    int rhs_mode = 0;
    const double* cpacc = nullptr;  // QpWs::cpacc
    const double* rbase = nullptr;  // QpWs::rbase
    double sigma_mu = 0;
};
template <int NK, int ROLE>
__device__ __forceinline__ void solve_staged(const QpDims& d, const QpWs& w, double* rhs, double* lds, SolveOut so = SolveOut{nullptr, nullptr, 0}) {
    using KS = KsLayout<NK>;
    constexpr int STG = KS::STG, SCH = KS::SCH, SM = KS::SM;
    const int tid = threadIdx.x, nj = d.nj, mid = twist_mid(nj);
    const int nl = mid, nr = nj - 1 - mid, SF = nl > nr ? nl : nr;
    const int nsteps = 2 * SF + 1;  // SF forward steps, the middle block, SF backward steps
    lds += KS::LEAD;                                   // (zero-filled with the stage buffers below)
    kl_lds* vec = (kl_lds*)(lds + QP_STAGE_BUFS * STG);  // nj*NK: rhs -> z -> x  (LDS pointers: through generic ones every access is a flat instruction)
    kl_lds* small = vec + ((nj * NK + 1) & ~1);        // [2 chains][A, Z, V][KS_VLEN], then [2 chains][64] partial sums
    if (so.rhs_mode) {  // rhs_from_acc into the LDS vector (same arithmetic): rhs = rbase + F'(G'v) of the sweep's accumulators
        constexpr int nu = NK / 3;
        const int oq = so.oq;
        const size_t ncp = (size_t)(NK / 9) * oq;
        const bool corrector = so.rhs_mode == 2;
        for (int it = tid; it < nj * nu; it += QP_THREADS) {
            const int j = it / nu + 1, u = it % nu, a = u / 3, k = u % 3;
            double g[6];  // G'v at control points 6(j-1)+3 .. 6j+2 of (agent a, dim k)
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const int j6 = 6 * (j - 1) + 3 + q;
                const __attribute__((address_space(1))) double* ac = QGC(so.cpacc) + (size_t)a * oq + j6;
                g[q] = corrector ? ac[k * ncp] - so.sigma_mu * ac[(3 + k) * ncp] : ac[(6 + k) * ncp];
            }
            const __attribute__((address_space(1))) double* L = QGC(so.Lk) + 9 * j;
            const size_t o0 = (size_t)(j - 1) * NK + u * 3;
#pragma unroll
            for (int e = 0; e < 3; ++e) vec[o0 + e] = QGC(so.rbase)[o0 + e] + g[3 + e] + L[0 + e] * g[0] + L[3 + e] * g[1] + L[6 + e] * g[2];
        }
    } else {  // rhs -> LDS: every thread's loads first, then its stores (one trip to memory instead of one per round; see apply_F)
        constexpr int RQ = 4;
        for (int i0 = tid; i0 < nj * NK; i0 += RQ * QP_THREADS) {
            double t[RQ];
#pragma unroll
            for (int q = 0; q < RQ; ++q) t[q] = i0 + q * QP_THREADS < nj * NK ? rhs[i0 + q * QP_THREADS] : 0.0;
#pragma unroll
            for (int q = 0; q < RQ; ++q)
                if (i0 + q * QP_THREADS < nj * NK) vec[i0 + q * QP_THREADS] = t[q];
        }
    }
    for (int i = tid; i < KS::VECS; i += QP_THREADS) small[i] = 0.0;
    // block indices handled at step s by the left / right wave (-1: idle)
    auto left_j = [&](int s) { return s < SF ? (s - (SF - nl) >= 0 ? s - (SF - nl) : -1) : (s == SF ? mid : (mid - 1 - (s - SF - 1) >= 0 ? mid - 1 - (s - SF - 1) : -1)); };
    auto right_j = [&](int s) { return s < SF ? (s - (SF - nr) >= 0 ? nj - 1 - (s - (SF - nr)) : -1) : (s == SF ? -1 : (mid + 1 + (s - SF - 1) <= nj - 1 ? mid + 1 + (s - SF - 1) : -1)); };
    // Staging (waves 2..): only the upper triangle of M_j and 1 / d_j are fetched -- NE elements per chain -- and every staging thread's
    // elements (the same for every step) are mapped once per call, so that a step is ONE batch of loads in flight per thread and a
    // batch of LDS stores.  What lies left of the diagonal in the stage buffers is zero-filled once per call.
    constexpr int NTRI = NK * (NK + 1) / 2, NE = NTRI + NK, NST = QP_THREADS - 128, MAXE = (2 * NE + NST - 1) / NST;
    int soff[MAXE], doff[MAXE];  // element u of this thread: source offset inside a knot's Lf slot, destination inside a stage (-1: none)
    if (ROLE == 1) {
#pragma unroll
        for (int u = 0; u < MAXE; ++u) {
            const int e = (tid - 128) + u * NST;
            soff[u] = 0, doff[u] = -1;
            if (e < 2 * NE) {
                const int side = e >= NE, idx = side ? e - NE : e;
                if (idx < NTRI) {
                    int rw = (int)(((2 * NK + 1) - sqrtf((float)((2 * NK + 1) * (2 * NK + 1) - 8 * idx))) * 0.5f);
                    if (rw * NK - rw * (rw - 1) / 2 > idx) rw--;
                    if ((rw + 1) * NK - (rw + 1) * rw / 2 <= idx) rw++;
                    const int k = rw + idx - (rw * NK - rw * (rw - 1) / 2);
                    soff[u] = rw * NK + k, doff[u] = side * SCH + rw * KL_LD + k;
                } else {
                    soff[u] = NK * NK + (idx - NTRI), doff[u] = side * SCH + SM + (idx - NTRI);
                }
            }
        }
    }
    for (int i = tid; i < KS::LEAD + QP_STAGE_BUFS * STG; i += QP_THREADS) (lds - KS::LEAD)[i] = 0.0;
    __syncthreads();
    auto stage = [&](int s, double* buf) {
        const int jl = left_j(s), jr = right_j(s);
        const double* srcl = w.Lf + (size_t)(jl >= 0 ? jl : 0) * KF_STRIDE(NK);
        const double* srcr = w.Lf + (size_t)(jr >= 0 ? jr : 0) * KF_STRIDE(NK);
        double tmp[MAXE];
#pragma unroll
        for (int u = 0; u < MAXE; ++u) {
            const bool right = doff[u] >= SCH;
            const bool on = doff[u] >= 0 && (right ? jr >= 0 : jl >= 0);
            tmp[u] = on ? (right ? srcr : srcl)[soff[u]] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < MAXE; ++u) {
            const bool right = doff[u] >= SCH;
            if (doff[u] >= 0 && (right ? jr >= 0 : jl >= 0)) buf[doff[u]] = tmp[u];
        }
    };
    // (round 6) the same in two halves, so that a staging thread's loads can stay in flight ACROSS step barriers: under full load a trip to global
    // memory (3-5 us) is longer than a chain step (~1.5 us), and a staging wave that loads, waits and stores inside one step made every
    // step as long as that trip however many stage buffers there were (which is why a third buffer never paid).  With QP_STAGE_REGS = R
    // register sets the loads of step s + QP_STAGE_BUFS - 1 + R are issued at step s and stored R steps later.
    auto stage_load = [&](int s, double (&tmp)[MAXE]) {
        if (s >= nsteps) return;
        const int jl = left_j(s), jr = right_j(s);
        // (explicitly GLOBAL loads and LDS stores: through generic pointers they are flat instructions, which count on both memory counters
        // and make the compiler wait for ALL of them at the first use -- no load would stay in flight across a barrier)
        typedef __attribute__((address_space(1))) const double gdbl;
        gdbl* srcl = (gdbl*)(w.Lf + (size_t)(jl >= 0 ? jl : 0) * KF_STRIDE(NK));
        gdbl* srcr = (gdbl*)(w.Lf + (size_t)(jr >= 0 ? jr : 0) * KF_STRIDE(NK));
        // no predicates: a load nobody needs reads a valid address (soff = 0, knot 0) and is never stored -- straight-line loads are what
        // lets the compiler count them (s_waitcnt vmcnt(n), n > 0) instead of waiting for all
#pragma unroll
        for (int u = 0; u < MAXE; ++u) tmp[u] = (doff[u] >= SCH ? srcr : srcl)[soff[u]];
    };
    auto stage_store = [&](int s, const double (&tmp)[MAXE], double* buf) {
        if (s >= nsteps) return;
        const int jl = left_j(s), jr = right_j(s);
        kl_lds* bl = (kl_lds*)buf;
#pragma unroll
        for (int u = 0; u < MAXE; ++u) {
            const bool right = doff[u] >= SCH;
            if (doff[u] >= 0 && (right ? jr >= 0 : jl >= 0)) bl[doff[u]] = tmp[u];
        }
    };
    if (ROLE == 1)
        for (int s0 = 0; s0 < QP_STAGE_BUFS - 1 && s0 < nsteps; ++s0) stage(s0, lds + s0 * STG);
    __syncthreads();
    const int wave = ROLE == 0 ? (tid >> 6) & 1 : 2, r = tid & 63;  // (role 1 only needs "wave >= 2")
    const bool act = r < NK;
    const int rr = act ? r : 0, g3 = 3 * (rr / 3), r3 = rr % 3;
    const int rs = act ? r : NK + 12;  // lanes >= NK write a slot behind the zero padding of the chain vectors
    if (wave < 2) __builtin_amdgcn_s_setprio(QP_CHAIN_PRIO);  // see twisted_factor
    // the three coupling coefficients a chain step starts with (forward: row rr of T_{j,jp}; backward: column rr of T_{jn,j}) come from
    // global memory: fetched ONE STEP AHEAD (the trip to the L2, ~1 us under load, was the first thing every step waited for)
    auto step_coef = [&](int s, double& q0, double& q1, double& q2) {
        q0 = q1 = q2 = 0.0;
        if (ROLE != 0 || s >= nsteps || s == SF) return;
        const int jb = wave == 0 ? left_j(s) : right_j(s);
        if (jb < 0) return;
        const int dir = wave == 0 ? +1 : -1;
        if (s < SF) {
            const bool has_prev = wave == 0 ? jb > 0 : jb + 1 < nj;
            if (has_prev) coupling_coef(w, jb - dir, dir, rr, q0, q1, q2);
        } else {
            const __attribute__((address_space(1))) double* E = QGC(w.Ek) + 9 * (dir > 0 ? jb + 1 : jb);
            q0 = dir > 0 ? E[3 * r3] : E[r3], q1 = dir > 0 ? E[3 * r3 + 1] : E[3 + r3], q2 = dir > 0 ? E[3 * r3 + 2] : E[6 + r3];
        }
    };
    double cf0, cf1, cf2;
    step_coef(0, cf0, cf1, cf2);
#ifdef QP_SOLVE_TIMERS  // developer build: left chain wave, SC 25 = work of the steps, 26 = waiting at the step barrier (staging wave: 29 / 30, right chain: 33 / 34)
    long long st_ = wall_clock64();
#define SOLVE_T(slot)                                                              \
    do {                                                                           \
        const long long t_ = wall_clock64();                                       \
        const int who_ = ROLE == 0 ? (tid == 0 ? 0 : (tid == 64 ? 8 : -1)) : (tid == 128 ? 4 : -1); /* left chain 25/26, staging wave 29/30, right chain 33/34 */ \
        if (w.prof && who_ >= 0) w.prof[slot + who_] += (double)(t_ - st_);        \
        st_ = t_;                                                                  \
    } while (0)
#else
#define SOLVE_T(slot)
#endif
    if (ROLE == 1 && QP_STAGE_REGS > 0) {
        constexpr int R = QP_STAGE_REGS > 0 ? QP_STAGE_REGS : 1;
        double tq[R][MAXE];
#pragma unroll
        for (int k = 0; k < R; ++k) stage_load(QP_STAGE_BUFS - 1 + k, tq[k]);
        // steady state, straight-line (no predicate, no branch between a load and its store: only then does the compiler count the loads in
        // flight instead of waiting for all of them): every element is stored -- the ones a thread does not have go to a padding column
        // nobody's products see (row 0, column KL_LD - 1, times the zero padding of the vectors), a chain that idles at a step gets knot 0's
        // numbers into its half of the buffer, which it does not read
        int dsto[MAXE];
#pragma unroll
        for (int u = 0; u < MAXE; ++u) dsto[u] = doff[u] >= 0 ? doff[u] : KL_LD - 1;
        typedef __attribute__((address_space(1))) const double gdbl;
        // The loads of the NEXT R steps are issued at the top of an iteration and first touched at its bottom, R step barriers later: the
        // compiler drains the memory counter completely wherever a loop-carried load is used (it does not count across a back edge), so the
        // one place where everything must have landed is made to be the place where it drains.
        double nq[R][MAXE];
        int s = 0;
        for (; s + 2 * R + QP_STAGE_BUFS - 1 <= nsteps; s += R) {
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int sl = s + R + k + QP_STAGE_BUFS - 1;
                const int jl = left_j(sl), jr = right_j(sl);
                gdbl* srcl = (gdbl*)(w.Lf + (size_t)(jl >= 0 ? jl : 0) * KF_STRIDE(NK));
                gdbl* srcr = (gdbl*)(w.Lf + (size_t)(jr >= 0 ? jr : 0) * KF_STRIDE(NK));
#pragma unroll
                for (int u = 0; u < MAXE; ++u) nq[k][u] = (doff[u] >= SCH ? srcr : srcl)[soff[u]];
            }
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int sp = s + k + QP_STAGE_BUFS - 1;
                kl_lds* bl = (kl_lds*)(lds + (sp % QP_STAGE_BUFS) * STG);
#pragma unroll
                for (int u = 0; u < MAXE; ++u) bl[dsto[u]] = tq[k][u];
                __syncthreads();
            }
#pragma unroll
            for (int k = 0; k < R; ++k)
#pragma unroll
                for (int u = 0; u < MAXE; ++u) tq[k][u] = nq[k][u];
        }
        for (; s < nsteps; s += R) {  // the last few steps (loads or stores that do not exist any more: predicated)
#pragma unroll
            for (int k = 0; k < R; ++k) {
                if (s + k < nsteps) {  // (uniform: every thread of the workgroup passes nsteps barriers, here or in the chains' loop)
                    const int sp = s + k + QP_STAGE_BUFS - 1;
                    stage_store(sp, tq[k], lds + (sp % QP_STAGE_BUFS) * STG);
                    stage_load(sp + R, tq[k]);
                    __syncthreads();
                }
            }
        }
    } else
    for (int s = 0; s < nsteps; ++s) {
        SOLVE_T(26);
        const double e0 = cf0, e1 = cf1, e2 = cf2;
        step_coef(s + 1, cf0, cf1, cf2);
        double* buf = lds + (s % QP_STAGE_BUFS) * STG;
        if (ROLE == 1) {
            const int sp = s + QP_STAGE_BUFS - 1;
            if (sp < nsteps) stage(sp, lds + (sp % QP_STAGE_BUFS) * STG);
        } else if (s == SF) {
            if (wave == 0) {  // middle block: forward with both neighbours' v, then backward; x_mid goes to both chains' V
                const kl_lds* Mst = (const kl_lds*)buf;
                kl_lds *A0 = (kl_lds*)small + KS_VPAD, *Z0 = A0 + KS_VLEN, *V0 = A0 + 2 * KS_VLEN, *V1 = A0 + 5 * KS_VLEN;
                kl_lds* ps = (kl_lds*)small + 6 * KS_VLEN;
                double v = vec[mid * NK + rr];
                if (mid > 0) {
                    double e0, e1, e2;
                    coupling_coef(w, mid - 1, +1, rr, e0, e1, e2);
                    v -= e0 * V0[g3] + e1 * V0[g3 + 1] + e2 * V0[g3 + 2];
                }
                if (mid + 1 < nj) {
                    double e0, e1, e2;
                    coupling_coef(w, mid + 1, -1, rr, e0, e1, e2);
                    v -= e0 * V1[g3] + e1 * V1[g3 + 1] + e2 * V1[g3 + 2];
                }
                A0[rs] = v;
                kl_sync();
                const double z = Mst[SM + rr] * kl_matvec_cols<NK>(Mst, A0, ps, r);
                Z0[rs] = z;
                kl_sync();
                const double x = kl_matvec_rows<NK>(Mst, Z0, ps, r);
                kl_sync();
                V0[rs] = x, V1[rs] = x;
                if (act) vec[mid * NK + r] = x;
            }
        } else {
            const bool fwd = s < SF;
            const int jb = wave == 0 ? left_j(s) : right_j(s);
            if (jb >= 0) {
                const kl_lds* Mst = (const kl_lds*)(buf + (wave == 0 ? 0 : SCH));
                kl_lds *A = (kl_lds*)small + KS_VPAD + (wave == 0 ? 0 : 3 * KS_VLEN), *Z = A + KS_VLEN, *V = A + 2 * KS_VLEN;
                kl_lds* ps = (kl_lds*)small + 6 * KS_VLEN + (wave == 0 ? 0 : 64);
                const double dinv = Mst[SM + rr];
                if (fwd) {
                    double v = vec[jb * NK + rr];
                    const bool has_prev = wave == 0 ? jb > 0 : jb + 1 < nj;
                    if (has_prev) v -= e0 * V[g3] + e1 * V[g3 + 1] + e2 * V[g3 + 2];  // r' = rhs_j - T_{j,jp} v_jp
                    kl_sync();
                    A[rs] = v;
                    kl_sync();
                    const double z = dinv * kl_matvec_cols<NK>(Mst, A, ps, r);
                    Z[rs] = z;
                    if (act) vec[jb * NK + r] = z;
                    kl_sync();
                    const double vj = kl_matvec_rows<NK>(Mst, Z, ps, r);
                    kl_sync();
                    V[rs] = vj;
                } else {
                    // w = T_{jn,j}' x_jn: column rr of the 3x3 block of its (agent, dim) group, rows g3 .. g3+2 (x_jn sits in V)
                    const double wv = e0 * V[g3] + e1 * V[g3 + 1] + e2 * V[g3 + 2];
                    kl_sync();
                    A[rs] = wv;
                    kl_sync();
                    const double u = vec[jb * NK + rr] - dinv * kl_matvec_cols<NK>(Mst, A, ps, r);
                    Z[rs] = u;
                    kl_sync();
                    const double x = kl_matvec_rows<NK>(Mst, Z, ps, r);
                    kl_sync();
                    V[rs] = x;
                    if (act) vec[jb * NK + r] = x;
                }
            }
        }
        SOLVE_T(25);
        __syncthreads();
    }
    if (wave < 2) __builtin_amdgcn_s_setprio(0);
    if (so.dx_out) {  // apply_F from the LDS vector (same arithmetic, same order)
        constexpr int nu = NK / 3;
        const int oq = so.oq;
        for (int it = tid; it < nj * nu; it += QP_THREADS) {
            const int j = it / nu + 1, u = it % nu;
            const kl_lds* uu = vec + (size_t)(j - 1) * NK + u * 3;
            const __attribute__((address_space(1))) double* L = QGC(so.Lk) + 9 * j;
            const double u0 = uu[0], u1 = uu[1], u2 = uu[2];
            __attribute__((address_space(1))) double* o = QG(so.dx_out) + (size_t)u * oq + 6 * (j - 1) + 3;  // control points 6(j-1)+3 .. 6j+2
            o[0] = L[0] * u0 + L[1] * u1 + L[2] * u2;
            o[1] = L[3] * u0 + L[4] * u1 + L[5] * u2;
            o[2] = L[6] * u0 + L[7] * u1 + L[8] * u2;
            o[3] = u0, o[4] = u1, o[5] = u2;
        }
        for (int it = tid; it < nu * 6; it += QP_THREADS) {  // the pinned ends
            const int u = it / 6, q = it % 6;
            QG(so.dx_out)[(size_t)u * oq + (q < 3 ? q : oq - 6 + q)] = 0.0;
        }
    } else {
        for (int i = tid; i < nj * NK; i += QP_THREADS) rhs[i] = vec[i];
    }
    __threadfence_block();
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------------------
// tiled path (nk > 36: batches of 5..64 agents, e.g. plan/batch_size = 8 or the joint QP of a whole mission).
// Knot blocks live in global memory (L2-resident), row-major with the leading dimension ld = nk rounded up to 16
// (padding rows/columns are identity), cut into 16x16 tiles for v_mfma_f64_16x16x4_f64:
//   factor, per knot j, LEFT-looking over tile columns p of the stacked panel [A_j ; C_j]  (A_j = T_jj - B B',
//   B = L_{j,j-1}, C_j = T_{j+1,j}):
//     (1) tile (ti,p) -= B(ti,:) B(p,:)' + A(ti,:16p) A(p,:16p)'      (8 waves, one MFMA tile each, K = ld + 16p)
//         C(ti,p)    -= C(ti,:16p) A(p,:16p)'
//     (2) every wave factors the 16x16 diagonal tile in registers (lane = row, v_readlane broadcasts) and
//     (3) solves its share of the rows below (one row per lane):  X <- X L_pp^{-T}
//   two workgroup barriers per tile column.  L_jj overwrites the lower triangle of Td[j], L_{j+1,j} overwrites To[j].
//   substitutions: block matvecs by the whole workgroup, the triangular solves by wave 0 tile by tile.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ d4 ld4(const double* p) { return *reinterpret_cast<const d4*>(p); }
// memory helpers of the tiled path: LDS = true turns the generic pointer into an LDS one (ds_read/ds_write instead of flat)
#define AS_LDS __attribute__((address_space(3)))
template <bool LDS>
__device__ __forceinline__ d4 ld4T(const double* p) {
    if (LDS) return *(const AS_LDS d4*)(p);
    return *reinterpret_cast<const d4*>(p);
}
template <bool LDS>
__device__ __forceinline__ double ldT(const double* p) {
    if (LDS) return *(const AS_LDS double*)(p);
    return *p;
}
template <bool LDS>
__device__ __forceinline__ void stT(double* p, double v) {
    if (LDS)
        *(AS_LDS double*)(p) = v;
    else
        *p = v;
}
template <bool LDS>
__device__ __forceinline__ void st4T(double* p, d4 v) {
    if (LDS)
        *(AS_LDS d4*)(p) = v;
    else
        *reinterpret_cast<d4*>(p) = v;
}

// acc += X(16 x K) Y(16 x K)'   (X, Y row-major, K a multiple of 16).  Lane (i = l&15, g = l>>4) loads X[i][k0+4g .. +3]:
// the four MFMA steps of a 16-chunk use k = 4g + s on both operands, a permutation of the summation index.
template <bool LDS>
__device__ __forceinline__ void tile_nt(d4& acc, const double* X, int ldx, const double* Y, int ldy, int K, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const double* xp = X + (size_t)i * ldx + 4 * g;
    const double* yp = Y + (size_t)i * ldy + 4 * g;
    for (int k0 = 0; k0 < K; k0 += 16) {
        const d4 xa = ld4T<LDS>(xp + k0), yb = ld4T<LDS>(yp + k0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[0], yb[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[1], yb[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[2], yb[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[3], yb[3], acc, 0, 0, 0);
    }
}

// C(16x16) -= acc   (C/D layout: row = (l>>4) + 4*reg, col = l&15)
template <bool LDS>
__device__ __forceinline__ void tile_sub(double* C, int ldc, const d4& acc, int lane) {
    const int i = lane & 15, g = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double* cp = C + (size_t)(g + 4 * r) * ldc + i;
        stT<LDS>(cp, ldT<LDS>(cp) - acc[r]);
    }
}

// global (ld x ld, dense) -> LDS (leading dimension ldl): all loads of a thread are issued before its LDS stores, so the
// copy costs one memory round trip, not one per element
__device__ __forceinline__ void stage_block(double* dst, int ldl, const double* src, int ld) {
    const int total = ld * ld;
    for (int base = threadIdx.x; base < total; base += 8 * QP_THREADS) {
        double tmp[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int it = base + u * QP_THREADS;
            tmp[u] = it < total ? src[it] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int it = base + u * QP_THREADS;
            if (it < total) *(AS_LDS double*)(dst + (it / ld) * ldl + it % ld) = tmp[u];
        }
    }
}

// one knot of the tiled factorisation: A (ld x ld, lower) <- chol(A - B B'), C <- C A^{-T}.  A, B, C may live in global
// memory or in LDS (generic pointers; leading dimension ld doubles, order 16*NT)
template <bool LDS>
__device__ __forceinline__ bool factor_knot_tiled(double* A, const double* B, double* C, int ld, int NT, int* flag) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    constexpr int NW = QP_THREADS / 64;
    const int n = 16 * NT;
    for (int p = 0; p < NT; ++p) {
        const int nA = NT - p, nC = (C && p > 0) ? NT : 0;
        if (B || p > 0) {
            for (int t = wave; t < nA + nC; t += NW) {
                d4 acc = d4{0, 0, 0, 0};
                const double* Ap = A + (size_t)p * 16 * ld;
                if (t < nA) {
                    const int ti = p + t;
                    if (B) tile_nt<LDS>(acc, B + (size_t)ti * 16 * ld, ld, B + (size_t)p * 16 * ld, ld, n, lane);
                    if (p > 0) tile_nt<LDS>(acc, A + (size_t)ti * 16 * ld, ld, Ap, ld, 16 * p, lane);
                    tile_sub<LDS>(A + (size_t)ti * 16 * ld + 16 * p, ld, acc, lane);
                } else {
                    const int ti = t - nA;
                    tile_nt<LDS>(acc, C + (size_t)ti * 16 * ld, ld, Ap, ld, 16 * p, lane);
                    tile_sub<LDS>(C + (size_t)ti * 16 * ld + 16 * p, ld, acc, lane);
                }
            }
            __threadfence_block();
            __syncthreads();
        }
        // diagonal tile: every wave factors its own register copy (lane&15 = row)
        double a[16];
        {
            const double* dp = A + (size_t)(16 * p + (lane & 15)) * ld + 16 * p;
            const d4 v0 = ld4T<LDS>(dp), v1 = ld4T<LDS>(dp + 4), v2 = ld4T<LDS>(dp + 8), v3 = ld4T<LDS>(dp + 12);
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] = v0[k], a[4 + k] = v1[k], a[8 + k] = v2[k], a[12 + k] = v3[k];
        }
        double inv;  // reciprocal of this lane's diagonal entry: the row solves below multiply instead of dividing
        const bool ok = chol_rows<16>(a, inv);
        if (!ok && tid == 0) *flag = 1;
        // rows below in A, all rows of C:  x <- x L_pp^{-T}
        const int nbelow = n - 16 * (p + 1), total = nbelow + (C ? n : 0);
        for (int base = 0; base < total; base += QP_THREADS) {
            const int idx = base + tid;
            const bool act = idx < total;
            double* rp = !act ? A : (idx < nbelow ? A + (size_t)(16 * (p + 1) + idx) * ld + 16 * p : C + (size_t)(idx - nbelow) * ld + 16 * p);
            double x[16];
            const d4 v0 = ld4T<LDS>(rp), v1 = ld4T<LDS>(rp + 4), v2 = ld4T<LDS>(rp + 8), v3 = ld4T<LDS>(rp + 12);
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k] = v0[k], x[4 + k] = v1[k], x[8 + k] = v2[k], x[12 + k] = v3[k];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                double sv = x[c];
#pragma unroll
                for (int k = 0; k < c; ++k) sv -= x[k] * rl(a[k], c);
                x[c] = sv * rl(inv, c);
            }
            if (act) {
                st4T<LDS>(rp, d4{x[0], x[1], x[2], x[3]});
                st4T<LDS>(rp + 4, d4{x[4], x[5], x[6], x[7]});
                st4T<LDS>(rp + 8, d4{x[8], x[9], x[10], x[11]});
                st4T<LDS>(rp + 12, d4{x[12], x[13], x[14], x[15]});
            }
        }
        __threadfence_block();
        __syncthreads();
        if (*flag) return false;
        if (wave == 0 && lane < 16) {  // nobody reads the diagonal tile again during the factorisation
            double* dp = A + (size_t)(16 * p + lane) * ld + 16 * p;
#pragma unroll
            for (int k = 0; k < 16; ++k) stT<LDS>(dp + k, k <= lane ? a[k] : 0.0);
        }
    }
    return true;
}

__device__ bool factor_tiled(const QpDims& d, const QpWs& w, int* flag, double* lds, int lds_avail) {
    const int ld = d.ldb, NT = ld / 16, tid = threadIdx.x;
    if (tid == 0) *flag = 0;
    __syncthreads();
    const int ldl = ld + 2;  // LDS leading dimension: rows 16 apart fall into different banks
    if (3 * ld * ldl <= lds_avail) {
        // LDS-resident variant (nk <= 72): the three blocks of a knot stay in LDS, every dependent step of the panel
        // loop is an LDS round trip instead of an L2 one.  T_{j+1,j} is generated in place from Ek (it is 3x3-block
        // diagonal), L_{j,j-1} is the previous knot's C buffer.
        double* bufA = lds;
        double* bufB = lds + (size_t)ld * ldl;
        double* bufC = lds + 2 * (size_t)ld * ldl;
        for (int j = 0; j < d.nj; ++j) {
            const double* Ag = w.Td + (size_t)j * ld * ld;
            stage_block(bufA, ldl, Ag, ld);
            const bool hasC = j + 1 < d.nj;
            if (hasC) {
                const double* E = w.Ek + 9 * (j + 1);
                for (int it = tid; it < ld * ld; it += QP_THREADS) {
                    const int rr = it / ld, cc = it % ld;
                    bufC[rr * ldl + cc] = (rr / 3 == cc / 3 && rr < d.nk && cc < d.nk) ? E[3 * (cc % 3) + (rr % 3)] : 0.0;
                }
            }
            __syncthreads();
            if (!factor_knot_tiled<true>(bufA, j > 0 ? bufB : nullptr, hasC ? bufC : nullptr, ldl, NT, flag)) return false;
            __syncthreads();  // the diagonal tiles are written late by wave 0
            double* Aw = w.Td + (size_t)j * ld * ld;
            for (int it = tid; it < ld * ld; it += QP_THREADS) Aw[it] = bufA[(it / ld) * ldl + it % ld];
            if (hasC) {
                double* Cw = w.To + (size_t)j * ld * ld;
                for (int it = tid; it < ld * ld; it += QP_THREADS) Cw[it] = bufC[(it / ld) * ldl + it % ld];
            }
            double* t = bufB;
            bufB = bufC, bufC = t;
            __syncthreads();
        }
    } else {
        for (int j = 0; j < d.nj; ++j) {
            double* A = w.Td + (size_t)j * ld * ld;
            const double* B = j > 0 ? w.To + (size_t)(j - 1) * ld * ld : nullptr;
            double* C = j + 1 < d.nj ? w.To + (size_t)j * ld * ld : nullptr;
            if (!factor_knot_tiled<false>(A, B, C, ld, NT, flag)) return false;
        }
    }
    __threadfence_block();
    __syncthreads();
    return true;
}

// v <- L^{-1} v for one knot block (wave 0; v in LDS, ld entries)
template <bool LDS>
__device__ __forceinline__ void trisolve_fwd(const double* L, int ld, int n, double* v, int lane) {
    const int NT = n / 16, i = lane & 15;
    for (int p = 0; p < NT; ++p) {
        double a[16];
        const double* dp = L + (size_t)(16 * p + i) * ld + 16 * p;
        const d4 v0 = ld4T<LDS>(dp), v1 = ld4T<LDS>(dp + 4), v2 = ld4T<LDS>(dp + 8), v3 = ld4T<LDS>(dp + 12);
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = v0[k], a[4 + k] = v1[k], a[8 + k] = v2[k], a[12 + k] = v3[k];
        double x = v[16 * p + i], dgl = 1.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) dgl = (i == k) ? a[k] : dgl;
        const double inv = 1.0 / dgl;  // one division per tile instead of one per dependent step
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const double xc = rl(x, c) * rl(inv, c);
            x = (i == c) ? xc : (i > c ? x - a[c] * xc : x);
        }
        if (lane < 16) v[16 * p + i] = x;
        for (int r0 = 16 * (p + 1); r0 < n; r0 += 64) {
            const int r = r0 + lane;
            const bool act = r < n;
            const double* rp = L + (size_t)(act ? r : 0) * ld + 16 * p;
            const d4 u0 = ld4T<LDS>(rp), u1 = ld4T<LDS>(rp + 4), u2 = ld4T<LDS>(rp + 8), u3 = ld4T<LDS>(rp + 12);
            double sv = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                sv += u0[k] * rl(x, k) + u1[k] * rl(x, 4 + k) + u2[k] * rl(x, 8 + k) + u3[k] * rl(x, 12 + k);
            if (act) v[r] -= sv;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// v <- L^{-T} v
template <bool LDS>
__device__ __forceinline__ void trisolve_bwd(const double* L, int ld, int n, double* v, int lane) {
    const int NT = n / 16, i = lane & 15;
    for (int p = NT - 1; p >= 0; --p) {
        double at[16];  // column i of the diagonal tile
#pragma unroll
        for (int k = 0; k < 16; ++k) at[k] = ldT<LDS>(L + (size_t)(16 * p + k) * ld + 16 * p + i);
        double x = v[16 * p + i], dgl = 1.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) dgl = (i == k) ? at[k] : dgl;
        const double inv = 1.0 / dgl;
#pragma unroll
        for (int c = 15; c >= 0; --c) {
            const double xc = rl(x, c) * rl(inv, c);
            x = (i == c) ? xc : (i < c ? x - at[c] * xc : x);
        }
        if (lane < 16) v[16 * p + i] = x;
        for (int k0 = 0; k0 < 16 * p; k0 += 64) {
            const int k = k0 + lane;
            const bool act = k < 16 * p;
            const double* cp = L + (size_t)(16 * p) * ld + (act ? k : 0);
            double sv = 0;
#pragma unroll
            for (int c = 0; c < 16; ++c) sv += ldT<LDS>(cp + (size_t)c * ld) * rl(x, c);
            if (act) v[k] -= sv;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// solve T du = rhs in place (rhs in global, nk per knot).  lds: 2*ld vector buffers + QP_THREADS partial sums.
__device__ void solve_tiled(const QpDims& d, const QpWs& w, double* rhs, double* lds, int lds_avail) {
    const int nk = d.nk, ld = d.ldb, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double* cur = lds;
    double* oth = lds + ld;
    double* part = lds + 2 * ld;
    // small blocks: the diagonal factor of the knot is staged in LDS (coalesced copy by all threads) so that the tile-by-
    // tile triangular solve of wave 0 is a chain of LDS round trips, not L2 ones
    const int ldl = ld + 2;
    double* Lst = part + QP_THREADS;
    const bool staged = 2 * ld + QP_THREADS + ld * ldl <= lds_avail;
    for (int j = 0; j < d.nj; ++j) {  // forward: v_j <- L_jj^{-1} (v_j - L_{j,j-1} v_{j-1})
        for (int r = tid; r < ld; r += QP_THREADS) cur[r] = r < nk ? rhs[(size_t)j * nk + r] : 0.0;
        __syncthreads();
        if (j > 0) {
            const double* Lo = w.To + (size_t)(j - 1) * ld * ld;
            const int i = tid & 15;
            for (int r0 = 0; r0 < nk; r0 += QP_THREADS / 16) {  // 16 lanes per row
                const int r = r0 + (tid >> 4);
                double sv = 0;
                if (r < nk)
                    for (int c = i; c < ld; c += 16) sv += Lo[(size_t)r * ld + c] * oth[c];
                sv += __shfl_xor(sv, 8), sv += __shfl_xor(sv, 4), sv += __shfl_xor(sv, 2), sv += __shfl_xor(sv, 1);
                if (r < nk && i == 0) cur[r] -= sv;
            }
            __syncthreads();
        }
        if (staged) {
            stage_block(Lst, ldl, w.Td + (size_t)j * ld * ld, ld);
            __syncthreads();
        }
        if (wave == 0) {
            if (staged)
                trisolve_fwd<true>(Lst, ldl, ld, cur, lane);
            else
                trisolve_fwd<false>(w.Td + (size_t)j * ld * ld, ld, ld, cur, lane);
        }
        __syncthreads();
        for (int r = tid; r < nk; r += QP_THREADS) rhs[(size_t)j * nk + r] = cur[r];
        double* t = cur;
        cur = oth, oth = t;
    }
    __threadfence_block();
    for (int j = d.nj - 1; j >= 0; --j) {  // backward: v_j <- L_jj^{-T} (v_j - L_{j+1,j}' v_{j+1})
        __syncthreads();
        for (int r = tid; r < ld; r += QP_THREADS) cur[r] = r < nk ? rhs[(size_t)j * nk + r] : 0.0;
        __syncthreads();
        if (j + 1 < d.nj) {
            const double* Lo = w.To + (size_t)j * ld * ld;
            if (ld <= QP_THREADS) {
                const int ng = QP_THREADS / ld, c = tid % ld, g = tid / ld;
                if (g < ng) {
                    double sv = 0;
                    for (int r = g; r < ld; r += ng) sv += Lo[(size_t)r * ld + c] * oth[r];
                    part[tid] = sv;
                }
                __syncthreads();
                if (tid < nk) {
                    double sv = 0;
                    for (int q = 0; q < ng; ++q) sv += part[q * ld + tid];
                    cur[tid] -= sv;
                }
            } else {
                for (int c = tid; c < nk; c += QP_THREADS) {
                    double sv = 0;
                    for (int r = 0; r < ld; ++r) sv += Lo[(size_t)r * ld + c] * oth[r];
                    cur[c] -= sv;
                }
            }
            __syncthreads();
        }
        if (staged) {
            stage_block(Lst, ldl, w.Td + (size_t)j * ld * ld, ld);
            __syncthreads();
        }
        if (wave == 0) {
            if (staged)
                trisolve_bwd<true>(Lst, ldl, ld, cur, lane);
            else
                trisolve_bwd<false>(w.Td + (size_t)j * ld * ld, ld, ld, cur, lane);
        }
        __syncthreads();
        for (int r = tid; r < nk; r += QP_THREADS) rhs[(size_t)j * nk + r] = cur[r];
        double* t = cur;
        cur = oth, oth = t;
    }
    __threadfence_block();
    __syncthreads();
}

// identity padding of the diagonal blocks (once per launch: the factorisation maps it onto itself)
__device__ void init_block_pads(const QpDims& d, const QpWs& w) {
    const int nk = d.nk, ld = d.ldb;
    if (ld == nk) return;
    const int npad = ld - nk;
    for (int it = threadIdx.x; it < d.nj * npad * ld; it += QP_THREADS) {
        const int j = it / (npad * ld), q = it % (npad * ld), r = nk + q / ld, c = q % ld;
        double* A = w.Td + (size_t)j * ld * ld;
        A[(size_t)r * ld + c] = r == c ? 1.0 : 0.0;
        A[(size_t)c * ld + r] = r == c ? 1.0 : 0.0;
    }
}

template <int ROLE>
__device__ __forceinline__ bool factor_dispatch(const QpDims& d, const QpWs& w, double* lA, int* flag, int lds_avail, const AsmArgs* asmb) {
    if (d.nk <= 36) {
        switch (d.nk) {
            case 9: return twisted_factor<9, ROLE>(d, w, flag, lA, asmb);
            case 18: return twisted_factor<18, ROLE>(d, w, flag, lA, asmb);
            case 27: return twisted_factor<27, ROLE>(d, w, flag, lA, asmb);
            default: return twisted_factor<36, ROLE>(d, w, flag, lA, asmb);
        }
    }
    if (ROLE != 0) return true;  // (the companion / assembling roles only exist on the wave path)
    return factor_tiled(d, w, flag, lA, lds_avail);
}

template <int ROLE>
__device__ __forceinline__ void solve_dispatch(const QpDims& d, const QpWs& w, double* rhs, double* lA, int lds_avail, SolveOut so = SolveOut{nullptr, nullptr, 0}) {
    if (d.nk <= 36) {
        switch (d.nk) {  // lA = start of the dynamic LDS region (the three block buffers are free between factorisations)
            case 9: solve_staged<9, ROLE>(d, w, rhs, lA, so); break;
            case 18: solve_staged<18, ROLE>(d, w, rhs, lA, so); break;
            case 27: solve_staged<27, ROLE>(d, w, rhs, lA, so); break;
            default: solve_staged<36, ROLE>(d, w, rhs, lA, so); break;
        }
        return;
    }
    if (ROLE != 1) solve_tiled(d, w, rhs, lA, lds_avail);  // (the staging role only exists on the wave path)
}


// The block factorisation / substitutions are compiled as stand-alone functions with by-value arguments: their register
// allocation (the wave-register path wants every VGPR) then neither depends on nor disturbs the row sweeps around them,
// and no kernel-level struct has its address taken.
struct BlkArgs {
    int nk, nj, ldb, lds_avail;
    double *Td, *To, *Lf;
    double* Ek;
    double* prof;
};
// Arguments of a non-kernel function arrive in VECTOR registers, and the compiler treats them as divergent: every pointer, dimension and
// address derived from them would live in VGPRs for the whole function -- that is what spills inside the knot
// loops, and a spill reload on a dependent chain costs a memory round trip.  They are wave-uniform by construction, so they are moved to
// scalar registers explicitly (v_readfirstlane); everything computed from them then stays scalar.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <class T>
__device__ __forceinline__ T* uni(T* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (T*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void blk_unpack(const BlkArgs& b, QpDims& d, QpWs& w) {
    d = QpDims{};
    w = QpWs{};
    d.nk = uni(b.nk), d.nj = uni(b.nj), d.ldb = uni(b.ldb), d.ld = d.nk + 1;
    w.Td = uni(b.Td), w.To = uni(b.To), w.Lf = uni(b.Lf), w.Ek = uni(b.Ek), w.prof = uni(b.prof);
}
__device__ __forceinline__ AsmArgs uni(const AsmArgs& A) {
    return AsmArgs{uni(A.cpacc), uni(A.pwgt), uni(A.Lk), uni(A.Dk), uni(A.normals), uni(A.Td), uni(A.N), uni(A.M), uni(A.nb), uni(A.first), uni(A.oq), uni(A.ldb)};
}
// (inlining these two was measured: 1100 VGPR spills, 31k instead of 51k agent-trajectories/s)
__device__ __noinline__ bool factor_entry_chain(BlkArgs b, AsmArgs A, double* lds, int* flag) {
    QpDims d;
    QpWs w;
    blk_unpack(b, d, w);
    const AsmArgs Au = uni(A);
    return factor_dispatch<0>(d, w, uni(lds), uni(flag), uni(b.lds_avail), &Au);
}
__device__ __noinline__ bool factor_entry_assemble(BlkArgs b, AsmArgs A, double* lds, int* flag) {
    QpDims d;
    QpWs w;
    blk_unpack(b, d, w);
    const AsmArgs Au = uni(A);
    return factor_dispatch<1>(d, w, uni(lds), uni(flag), uni(b.lds_avail), &Au);
}
__device__ __noinline__ bool factor_entry_follow(BlkArgs b, AsmArgs A, double* lds, int* flag) {
    QpDims d;
    QpWs w;
    blk_unpack(b, d, w);
    const AsmArgs Au = uni(A);
    return factor_dispatch<2>(d, w, uni(lds), uni(flag), uni(b.lds_avail), &Au);
}
__device__ __forceinline__ bool factor_entry(const BlkArgs& b, const AsmArgs& A, double* lds, int* flag) {
    const int wave = threadIdx.x >> 6;
    if (b.nk <= 36 && wave >= 4) return factor_entry_assemble(b, A, lds, flag);
    if (b.nk <= 36 && wave >= 2) return factor_entry_follow(b, A, lds, flag);
    return factor_entry_chain(b, A, lds, flag);
}
__device__ __noinline__ void solve_entry_chain(BlkArgs b, double* rhs, double* lds, SolveOut so) {
    QpDims d;
    QpWs w;
    blk_unpack(b, d, w);
    solve_dispatch<0>(d, w, uni(rhs), uni(lds), uni(b.lds_avail), SolveOut{uni(so.dx_out), uni(so.Lk), uni(so.oq), uni(so.rhs_mode), uni(so.cpacc), uni(so.rbase), so.sigma_mu});
}
__device__ __noinline__ void solve_entry_stage(BlkArgs b, double* rhs, double* lds, SolveOut so) {
    QpDims d;
    QpWs w;
    blk_unpack(b, d, w);
    solve_dispatch<1>(d, w, uni(rhs), uni(lds), uni(b.lds_avail), SolveOut{uni(so.dx_out), uni(so.Lk), uni(so.oq), uni(so.rhs_mode), uni(so.cpacc), uni(so.rbase), so.sigma_mu});
}
// (every wave passes the same workgroup barriers in either function)
__device__ __forceinline__ void solve_entry(const BlkArgs& b, double* rhs, double* lds, SolveOut so = SolveOut{nullptr, nullptr, 0}) {
    if (b.nk <= 36 && (threadIdx.x >> 6) >= 2)
        solve_entry_stage(b, rhs, lds, so);
    else
        solve_entry_chain(b, rhs, lds, so);
}

#define QP_POLISH_PART 2
#include "qp_polish.inc"
#undef QP_POLISH_PART

// ------------------------------------------------------------------------------------------------------------
// build_dummy (rbp_planner.hpp:513-549): control points of the waypoint-constant trajectory.  Used as `dummy`
// in sequential mode and as the interior-point warm start in every mode.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dummy_kernel(DevSession s) {
    const int MS = s.M, PS = MS + 1, oqS = 6 * MS;  // slot strides (the session's largest M)
    const size_t per_mission = (size_t)s.N * 3 * oqS, total = (size_t)s.K * per_mission;
    for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (size_t)gridDim.x * blockDim.x) {
        const int mission = (int)(it / per_mission);
        const int M = s.Mk[mission], P = M + 1, oq = 6 * M;
        const size_t rest = it % per_mission;
        const int j6 = (int)(rest % oqS), k = (int)((rest / oqS) % 3), qi = (int)(rest / ((size_t)3 * oqS));
        if (j6 >= oq) continue;
        const int m = j6 / 6, j = j6 % 6;
        // idx runs with m (one waypoint pair per segment); the `idx >= size-1` branch is unreachable for M+1 waypoints
        const float* tr = s.init_traj + (size_t)mission * s.N * PS * 3 + (size_t)qi * P * 3;
        const int a = (j < 3) ? 0 : 1;
        s.ctrl[(size_t)mission * per_mission + ((size_t)qi * 3 + k) * oq + j6] = (1 - a) * (double)tr[3 * m + k] + a * (double)tr[3 * (m + 1) + k];
    }
}

#ifdef QP_TRACE
#ifndef QP_TRACE_BATCH
#define QP_TRACE_BATCH 0
#endif
// developer build: per-iteration checksums of the first batch QP of a mission, written into the mission's coef slot
#define TRC(slot, expr)                                                         \
    do {                                                                        \
        if (batch == QP_TRACE_BATCH && pass_index == 0 && iter < 100) {                       \
            const double v_ = (expr);                                           \
            if (tid == 0) trc[iter * 16 + (slot)] = v_;                         \
        }                                                                       \
    } while (0)
__device__ double trc_sum(const double* p, size_t n, double* red) {
    double a = 0;
    for (size_t i = threadIdx.x; i < n; i += QP_THREADS) a += fabs(p[i]);
    return block_reduce(a, 0, red);
}
#else
#define TRC(slot, expr)
#endif

// ------------------------------------------------------------------------------------------------------------
// set-up of one batch QP (all threads of the workgroup): mission constants, the SFC box of every (batch agent, segment), the pinned end
// control points, the presolve lists and the row constants.  false: the mission was abandoned (S.status set).
// ------------------------------------------------------------------------------------------------------------
// far_R: QP_FAR_SLACK of this attempt (1e300: every row near); nfar: how many (group, neighbour) entries were classified far.
__device__ __forceinline__ bool qp_setup_batch(const DevSession& S, RowCtx& c, double* ctrl, const double* T, int mission, int first, int nb,
                                               int* flag, double* lds, int& frozen_free_rows, double far_R, int& nfar) {
    const int tid = threadIdx.x, N = S.N;
    const QpDims& d = c.d;
    const QpWs& w = c.w;
    const int M = d.M;
    mission_constants(d, T, const_cast<QpWs&>(w));
    init_block_pads(d, w);

    // SFC box of every (batch agent, segment): first box with end time >= T[m+1]  (rbp_planner.hpp:447-453)
    // (the box end times go through LDS: the scan below is a chain of dependent reads -- four threads, ~40 steps each --, and from
    // global memory every step was a trip of its own; the boxes themselves are then copied by all threads at once)
    {
        const int need = nb * S.max_boxes + (M + 2) + (nb * M + 1) / 2 + 1;
        const bool stage = need <= c.lds_avail;          // (wide batches of the tiled path: straight from global memory)
        double* bt_l = lds;                              // [nb][max_boxes]
        double* T_l = lds + nb * S.max_boxes;            // [M + 1]
        int* sel_l = stage ? (int*)(T_l + M + 2) : w.fcnt;  // [nb][M]  (fcnt is filled by the presolve below)
        if (stage) {
            for (int it = tid; it < nb * S.max_boxes; it += QP_THREADS)
                bt_l[it] = S.sfc_time[((size_t)mission * N + first + it / S.max_boxes) * S.max_boxes + it % S.max_boxes];
            for (int it = tid; it <= M; it += QP_THREADS) T_l[it] = T[it];
            __syncthreads();
        }
        for (int a = tid; a < nb; a += QP_THREADS) {
            const int qa = first + a;
            int nbx = S.sfc_count[(size_t)mission * N + qa];
            if (nbx <= 0) {  // no SFC boxes: the planner stage was run on a plan whose corridor was never computed (caller error)
                atomicCAS(&S.status[mission], 0, (int)RBP_ERR_BAD_ARGUMENT);
                nbx = 1;  // keeps the reads below inside the agent's slot; the mission is abandoned after the barrier
            }
            const double* bt = stage ? bt_l + a * S.max_boxes : S.sfc_time + ((size_t)mission * N + qa) * S.max_boxes;
            const double* Tt = stage ? T_l : T;
            int bi = 0;
            for (int m = 0; m < M; ++m) {
                while (bi < nbx && bt[bi] < Tt[m + 1]) bi++;
                sel_l[a * M + m] = bi < nbx ? bi : nbx - 1;
            }
        }
        __threadfence_block();
        __syncthreads();
        for (int it = tid; it < nb * M * 3; it += QP_THREADS) {
            const int am = it / 3, k = it % 3, a = am / M;
            const double* bx = S.sfc_box + ((size_t)mission * N + first + a) * S.max_boxes * 6 + 6 * sel_l[am];
            w.boxlo[(size_t)am * 3 + k] = bx[k];
            w.boxhi[(size_t)am * 3 + k] = bx[3 + k];
        }
        __threadfence_block();
        __syncthreads();
    }
    // pin the six end control points of the batch agents to the start/goal state (rows 0-5 of Aeq_base, :380-387)
    for (int it = tid; it < nb * 3; it += QP_THREADS) {
        const int a = it / 3, k = it % 3, qa = first + a;
        const double* st = S.start + ((size_t)mission * N + qa) * 9;
        const double* gl = S.goal + ((size_t)mission * N + qa) * 9;
        const double h0 = T[1] - T[0], hT = T[M] - T[M - 1];
        double* x = ctrl + ((size_t)qa * 3 + k) * d.oq;
        x[0] = st[k], x[1] = x[0] + h0 * st[k + 3] / 5, x[2] = 2 * x[1] - x[0] + h0 * h0 * st[k + 6] / 20;
        double* xe = x + 6 * (M - 1);
        xe[5] = gl[k], xe[4] = xe[5] - hT * gl[k + 3] / 5, xe[3] = 2 * xe[4] - xe[5] + hT * hT * gl[k + 6] / 20;
    }
    __threadfence();
    __syncthreads();
    if (__hip_atomic_load(&S.status[mission], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;  // set above: an agent without boxes

    // ---- presolve: a frozen-neighbour row  sg*n.(d_f - x_a) >= r_a + r_f  is implied by the SFC bounds of (a, segment) when
    // its slack is positive for EVERY x_a in the box; such rows cannot be active and are dropped (exact: the feasible
    // set is unchanged).  Per (agent, segment) the surviving neighbours are listed; ~72 % of the rows go away on the
    // 64-agent missions.
    for (int it = tid; it < nb * M; it += QP_THREADS) {
        const int a = it / M, seg = it % M, qa = first + a;
        const double ra = c.radius[qa];
        const double* lo = w.boxlo + ((size_t)a * M + seg) * 3;
        const double* hi = w.boxhi + ((size_t)a * M + seg) * 3;
        int cnt = 0, nfar_g = 0;
        int* fl = w.flist + (size_t)it * N;
        double xa6[6][3];  // this group's own control points at the starting point (the near / far classification, see QP_FAR_SLACK)
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int e = 0; e < 3; ++e) xa6[i][e] = ctrl[((size_t)qa * 3 + e) * d.oq + 6 * seg + i];
        // (four neighbours per round, all their loads ahead of the list stores: the compiler does not move a load across a store, and a
        // neighbour per round meant 60 trips to memory in a row, ~1 us each under load)
        for (int f0 = 0; f0 < N && N > 1; f0 += 4) {
            bool keep4[4], far4[4];
            float nvq[4][3];
            double cf[4][6][3];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int f = f0 + q < N ? f0 + q : N - 1;
                const bool a_first = qa < f;
                const int lo_ = a_first ? qa : f, hi_ = a_first ? f : qa;
                const float* nv = c.normals + (pair_index(N, lo_ == hi_ ? 0 : lo_, lo_ == hi_ ? 1 : hi_) * M + seg) * 3;
#pragma unroll
                for (int e = 0; e < 3; ++e) nvq[q][e] = nv[e];
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int e = 0; e < 3; ++e) cf[q][i][e] = ctrl[((size_t)f * 3 + e) * d.oq + 6 * seg + i];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int f = f0 + q;
                const bool a_first = qa < f;
                const double sg = a_first ? 1.0 : -1.0;
                const double n0 = sg * (double)nvq[q][0], n1 = sg * (double)nvq[q][1], n2 = sg * (double)nvq[q][2];
                const double mx = fmax(n0 * lo[0], n0 * hi[0]) + fmax(n1 * lo[1], n1 * hi[1]) + fmax(n2 * lo[2], n2 * hi[2]);
                const double rr = ra + c.radius[f < N ? f : 0];
                bool keep = false;
                double smin = 1e300;  // smallest slack of the group's six rows at the starting point
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const double nd = n0 * cf[q][i][0] + n1 * cf[q][i][1] + n2 * cf[q][i][2];
                    if (!(nd - rr - mx > 1e-6)) keep = true;
                    smin = fmin(smin, nd - rr - (n0 * xa6[i][0] + n1 * xa6[i][1] + n2 * xa6[i][2]));
                }
                keep4[q] = keep && f < N && !(f >= first && f < first + nb);
                far4[q] = smin > far_R;
            }
            // the near neighbours fill the group's list from the front, the far ones from the back (entry idx >= near count of the group's
            // rows is fl[N - 1 - (idx - near count)]): the rows of the interior-point phase come first in every column
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (keep4[q]) {
                    if (far4[q])
                        fl[N - 1 - nfar_g++] = f0 + q;
                    else
                        fl[cnt++] = f0 + q;
                }
        }
        w.fnear[it] = cnt;
        w.fcnt[it] = cnt + nfar_g;
    }
    __threadfence_block();
    __syncthreads();
    // offsets of the groups' normal lists (exclusive prefix sum of the counts) and the number of non-constant frozen rows: every group
    // sums its predecessors itself -- loads only, all in flight together; one thread walking the 144 groups with a store per step made
    // 144 trips to memory in a row
    if (tid == 0) flag[0] = 0, flag[1] = 0;
    __syncthreads();
    for (int it = tid; it < nb * M; it += QP_THREADS) {
        int acc = 0;
        for (int o = 0; o < it; ++o) acc += w.fcnt[o];
        w.fbase[it] = acc;
        const int seg = it % M;
        atomicAdd(flag, w.fnear[it] * ((seg == 0 || seg == M - 1) ? (M == 1 ? 0 : 3) : 6));  // (rows of the interior-point phase: the near ones)
        atomicAdd(flag + 1, w.fcnt[it] - w.fnear[it]);
    }
    __threadfence_block();
    __syncthreads();
    frozen_free_rows = *flag;
    nfar = flag[1];
    __syncthreads();
    // sweep work order: the threads of a wavefront walk the row lists of their control points in lockstep, so a wave
    // costs as much as its longest list.  Handing out the (agent, segment) groups by falling row count puts lists of
    // similar length into the same wave (and the long ones into the first round).  Rank sort, one thread per group.
    for (int it = tid; it < nb * M; it += QP_THREADS) {
        const int ci = w.fcnt[it];
        int rank = 0;
        for (int o = 0; o < nb * M; ++o) {
            const int co = w.fcnt[o];
            rank += (co > ci || (co == ci && o < it)) ? 1 : 0;
        }
        w.fperm[rank] = it;
        w.frank[it] = rank;
    }
    __threadfence_block();
    __syncthreads();
    // row storage (see the note above QpWs): tile t holds the control points wi = 64 t .. 64 t + 63; its columns are as long as
    // its first (= longest) one
    for (int t = tid; t <= d.ntile; t += QP_THREADS) {  // (every tile sums its predecessors: see fbase above)
        int base = 0;
        for (int o = 0; o < t; ++o) base += 64 * (d.ncol0 + w.fcnt[w.fperm[(64 * o) / 6]]);
        w.tile_base[t] = base;
    }
    for (int wi = tid; wi < nb * d.oq; wi += QP_THREADS) {
        const int grp = w.fperm[wi / 6], a = grp / M, seg = grp - a * M;
        w.wi_of[a * d.oq + 6 * seg + wi % 6] = wi;
    }
    __threadfence_block();
    __syncthreads();
    // tabulate the constants of the surviving frozen rows: the signed normal per (group, neighbour) and, per row,
    // rh = n . d_f - (r_a + r_f)
    for (int it = tid; it < nb * M * 6; it += QP_THREADS) {
        const int as = it / 6, i = it % 6, a = as / M, seg = as % M, qa = first + a, j6 = 6 * seg + i;
        const int cnt = w.fcnt[as], cn = w.fnear[as];
        const int* fl = w.flist + (size_t)as * N;
        const int wi = 6 * w.frank[as] + i;
        const size_t r0 = (size_t)w.tile_base[wi >> 6] + (wi & 63) + (size_t)d.ncol0 * 64;
        float* nr = w.nrm + (size_t)w.fbase[as] * 3;
        const double ra = c.radius[qa];
        for (int i0 = 0; i0 < cnt; i0 += 4) {  // (four rows per round, loads first: see the presolve loop)
            int fq[4];
            float nvq[4][3];
            double cfq[4][3];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = i0 + q < cnt ? i0 + q : cnt - 1;
                fq[q] = fl[e < cn ? e : N - 1 - (e - cn)];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int f = fq[q];
                const bool a_first = qa < f;
                const float* nv = c.normals + (pair_index(N, a_first ? qa : f, a_first ? f : qa) * M + seg) * 3;
#pragma unroll
                for (int e = 0; e < 3; ++e) nvq[q][e] = nv[e], cfq[q][e] = ctrl[((size_t)f * 3 + e) * d.oq + j6];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = i0 + q, f = fq[q];
                if (idx < cnt) {
                    const float sgf = qa < f ? 1.0f : -1.0f;
                    const double n0 = (double)(sgf * nvq[q][0]), n1 = (double)(sgf * nvq[q][1]), n2 = (double)(sgf * nvq[q][2]);
                    if (i == 0) nr[3 * idx] = sgf * nvq[q][0], nr[3 * idx + 1] = sgf * nvq[q][1], nr[3 * idx + 2] = sgf * nvq[q][2];
                    w.rh[r0 + (size_t)idx * 64] = n0 * cfq[q][0] + n1 * cfq[q][1] + n2 * cfq[q][2] - (ra + c.radius[f]);
                }
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    return true;
}

// ------------------------------------------------------------------------------------------------------------
// one batch QP of one mission (all threads of the workgroup)
// ------------------------------------------------------------------------------------------------------------
// far_R: see QP_FAR_SLACK.  Returns 1 when the batch QP has to be solved again with every row near (the caller puts the batch agents'
// control points back to the starting point first), else 0.
__device__ __forceinline__ int qp_batch_body(const DevSession& S, double* ws_base, size_t ws_stride, int mission, int batch, int nbmax,
                                             int reset_cost, int lds_doubles, int pass_index, double far_R) {
    const int tid = threadIdx.x;
    if (S.status[mission] != 0) return 0;
    // M of this mission: made wave-uniform explicitly (an SGPR like every other dimension; as a per-lane value it was spilled
    // and reloaded under divergent control flow with some lanes reading garbage)
    const int N = S.N, M = __builtin_amdgcn_readfirstlane(S.Mk[mission]), MS = S.M;  // MS: slot stride of the per-mission arrays
    const int first = batch * nbmax;
    const int nb = min(nbmax, N - first);
    if (nb <= 0) return 0;
    RowCtx c;
    c.meta = nullptr;
    c.scal = S.scalars + (size_t)mission * SC_N, c.mission = mission, c.lds_avail = lds_doubles - 32;
    c.d = make_dims(N, M, first, nb);
    c.w = carve(ws_base + (size_t)mission * ws_stride, c.d, nbmax);
    double* ctrl = S.ctrl + (size_t)mission * N * 3 * 6 * MS;
    c.ctrl = ctrl;
    c.normals = S.rsfc_normal + (size_t)mission * S.npair * MS * 3;
    c.radius = S.radius + (size_t)mission * N;
    const QpDims& d = c.d;
    const QpWs& w = c.w;
    const double* T = S.T + (size_t)mission * (MS + 1);
    double* scal = S.scalars + (size_t)mission * SC_N;

    // dynamic LDS: [0,32) reduction scratch + flags (always live), then a work area shared in turn by the block
    // factorisation (3 blocks), the staged substitutions (4 padded blocks + rhs) and the polish
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
    double* red = lds_raw;  // 16
    int* flag = (int*)(lds_raw + 16);
    double* lds = lds_raw + 32;
    double* lA = lds;
#ifdef QP_PROFILE
    const BlkArgs ba{d.nk, d.nj, d.ldb, c.lds_avail, w.Td, w.To, w.Lf, w.Ek, scal};
#else
    const BlkArgs ba{d.nk, d.nj, d.ldb, c.lds_avail, w.Td, w.To, w.Lf, w.Ek, nullptr};
#endif
    double* red2 = red;
    int* flag2 = flag;

    PROF_DECL;
    int frozen_free_rows = 0, nfar = 0;
    if (!qp_setup_batch(S, c, ctrl, T, mission, first, nb, flag, lds, frozen_free_rows, far_R, nfar)) return 0;
    PROF(8);  // batch setup: constants, SFC boxes, presolve lists, row constants
    PassIO io;
    __shared__ RowCtx c_lds;  // the sweeps' view of the row context (see sweep<>)
#if QP_ROW_BLK > 0
    __shared__ SweepMeta meta_lds;
    const bool meta_ok = nb * d.M <= QP_META_G && d.ntile + 1 <= 16 && d.N <= 256 && (size_t)nb * d.M * d.N < 65536;
    __syncthreads();
    if (meta_ok) {
        for (int rk = tid; rk < nb * d.M; rk += QP_THREADS) {
            const int g = w.fperm[rk];
            meta_lds.grp[rk] = (unsigned char)g, meta_lds.near[rk] = (unsigned char)w.fnear[g], meta_lds.all[rk] = (unsigned char)w.fcnt[g];
            meta_lds.fbase[rk] = (unsigned short)w.fbase[g];
        }
        for (int t = tid; t <= d.ntile; t += QP_THREADS) meta_lds.tile_base[t] = w.tile_base[t];
    }
    c.meta = meta_ok ? &meta_lds : nullptr;
#endif
    __syncthreads();
    if (tid == 0) c_lds = c;
    __syncthreads();
    io.mu0 = QP_MU0, io.s_floor = QP_SFLOOR, io.dreg = 1e-9, io.sigma_mu = 0, io.alpha = 0;
    // presolve: constant rows (pinned control points) must hold within 1e-6 (CPLEX default feasibility tolerance)
    io.vmax = 0;
    row_pass<PASS_PRESOLVE>(c, io);
    const double pin_viol = block_reduce(io.vmax, 1, red);
    if (pin_viol > 1e-6) {
        if (tid == 0) {
            atomicCAS(&S.status[mission], 0, (int)RBP_ERR_QP_FAILED);
#ifndef QP_PROFILE
            scal[SC_PROF0 + 1] = 1000.0 * batch + 1, scal[SC_PROF0 + 2] = pin_viol;  // why: a constant (pinned) row is violated
#endif
        }
        return 0;
    }
    const double nrows_free = (double)((size_t)(d.oq - 6) * (6 * d.nb + d.npb)) + (double)frozen_free_rows;  // every pair row once
    bool ok = false;
    int it_count = 0, polished = 0, early_tries = 0, fail_reason = 3;
    double gap_next = 0, pres_next = 0, kkt_ipm = 0;
    double flops = 0, rows_swept = 0, row_bytes = 0, sweep_bytes = 0;
    // ALGORITHMIC HBM bytes of one interior-point iteration (the numerator of the HBM roofline): per row the three
    // sweeps read (s, z) three times and write them once (64 B), frozen rows also read their constant three times (+24 B); per free
    // control point the accumulators are written twice and read four times (288 B); the knot blocks: on the tiled path 8 block
    // transfers of ldb^2 doubles per knot (T_j written and read, two factor blocks written once and read by both substitutions); on the
    // wave path the knot's factor is ONE triangle, M_j = L_j^-T, written once and staged four times (forward and backward pass of the two
    // substitutions), plus the reciprocal pivots: 5 triangles + 5 nk.  (Until round 6 the model also counted T_j written and read as a
    // triangle -- 7 triangles -- which both builds have stopped doing: the blocks are assembled just in time into LDS images and never
    // reach global memory, so those bytes are no longer part of what the algorithm has to move.)
    const double blk_doubles = d.nk <= 36 ? 5.0 * (d.nk * (d.nk + 1) / 2) + 5.0 * d.nk : 8.0 * d.ldb * d.ldb;
    const double bytes_sweeps = 88.0 * frozen_free_rows + 64.0 * (double)(d.oq - 6) * (6.0 * d.nb + 2.0 * d.npb) + 288.0 * (double)d.nb * (d.oq - 6);
    const double bytes_iter = bytes_sweeps + 8.0 * (double)d.nj * blk_doubles;
    PolishWs pw;
    pw.cand = (Cand*)w.polish;
    pw.V = w.polish + PL_NC * 14;
    pw.Sg = pw.V + (size_t)(PL_NC + 1) * d.nj * d.nk;
    pw.ncand = (int*)(pw.Sg + PL_NC * PL_NC);
    pw.Lbig = pw.Sg + PL_NC * PL_NC + 8;
    row_pass<PASS_INIT>(c, io);
    __threadfence_block();
    __syncthreads();
#ifdef QP_TRACE
    double* trc = S.coef + (size_t)mission * N * 3 * 6 * MS;
#endif
    for (int iter = 0; iter < QP_MAX_ITERS; ++iter) {
        it_count = iter;
        // ---- sweep 1: weights, accumulators, residual norms (from the second iteration on it is fused into the previous
        // iteration's update sweep)
        PROF(0);
        if (iter == 0) {
            io.sum0 = 0, io.vmax = 0;
            SWEEP(PASS_BUILD);
        }
        PROF(1);
        const double gap = iter == 0 ? block_reduce(io.sum0, 0, red) : gap_next;
        const double pres = iter == 0 ? block_reduce(io.vmax, 1, red) : pres_next;
        __threadfence_block();
        __syncthreads();
        double dmax, gmax;
        rbase_from_acc(c, dmax, gmax);  // rbase = -F'(2Qx + G'z)
        const double dres = block_reduce(dmax, 1, red) / (1.0 + block_reduce(gmax, 1, red));
        const double mu = gap / nrows_free;
        rows_swept += nrows_free;
        kkt_ipm = fmax(pres, fmax(dres, mu));
        TRC(0, gap); TRC(1, pres); TRC(2, dres); TRC(3, trc_sum(w.rbase, (size_t)d.nj * d.nk, red)); TRC(4, trc_sum(w.cpacc, (size_t)12 * d.nb * d.oq, red));
        if (pres < 1e-9 && dres < 1e-9 && mu < 1e-10) {
            ok = true;
            break;
        }
        // rows inconsistent at rounding level (no interior and infeasible by ~1e-9): the regularised iteration settles
        // on the least-violation point with pres stuck; accept below CPLEX's default feasibility tolerance 1e-6.
        if (pres < 1e-6 && dres < 1e-9 && mu < 1e-13) {
            ok = true;
            break;
        }
        // complementarity has collapsed (mu < 1e-14) with the primal residual at rounding level while the dual residual sits on
        // its noise floor (Newton weights of 1e9+ on degenerate active sets: ~5e-9 relative): more iterations change nothing.
        // The active-set polish below turns this point into the certified optimum; if it is refused, kkt_max reports the floor.
        if (pres < 1e-9 && dres < 1e-7 && mu < 1e-14) {
            ok = true;
            break;
        }
        PROF(2);
        // EARLY CROSSOVER: the polish returns the exact optimum (KKT-verified on every row) as soon as the interior-point
        // iterate identifies the active set, which happens several iterations before the 1e-10 termination test: try it
        // when the residuals and mu fall below QP_EARLY_TOL = 1e-6 and once more at mu < 1e-8 (tuned on the 50-map sweep);
        // a refused attempt leaves the iterate untouched and the loop goes on to the 1e-10 test and the final polish.
        // POLISH FIRST (Gauss-Seidel passes >= 2, e.g. plan/iteration = 50 of BASELINE config C5): the current point is the
        // previous pass's optimum of this batch, only the frozen neighbours have moved a little, so the active set is
        // mostly unchanged.  Before the first interior-point iteration the active-set step is tried directly with the rows
        // near their bounds as candidates; it is accepted only under the same full KKT check as always (so it IS the
        // optimum of this pass's QP), otherwise the interior-point method runs as in the first pass.
        const bool polish_first = QP_POLISH_FIRST && pass_index > 0 && iter == 0;
        const bool early = QP_EARLY_POLISH && early_tries < QP_EARLY_TRIES && pres < QP_EARLY_TOL && dres < QP_EARLY_TOL &&
                           mu < (early_tries == 0 ? QP_EARLY_TOL : 1e-2 * QP_EARLY_TOL);
        if (S.p.polish && (polish_first || early)) {
            if (!polish_first) early_tries++;
            const int acc = polish_entry(c, pw, lds, red2, flag2, polish_first ? 1 : 0);
            __syncthreads();
            PROF(0);
#ifdef QP_POLSTATS
            if (tid == 0 && !polish_first) scal[25] += (early_tries == 1), scal[26] += (early_tries == 1 && acc == 0), scal[27] += (early_tries == 2), scal[29] += (early_tries == 2 && acc == 0);  // (tools/experiments/r05_polstats.py)
#endif
            if (acc == 0) {
                ok = true, polished = 1;
                break;
            }
        }
        // ---- Newton matrix and factorisation
        // wave path, 512 threads: assembled behind the factorisation chains by waves 4.. (twisted_factor); the 256-thread build has no
        // waves to spare (two chains, two companions) and assembles up front with all of them
        if (d.nk > 36)
            assemble_blocks(c, lds);
        else if (ASM_HELPERS == 0 && !QP_FOLLOW_ASM)
            assemble_blocks_lds(asm_args(c), d.nj, lds);  // (256-thread build, round 6: the chains' companion waves assemble just in time instead)  // (256-thread build, round 6: the chains' companion waves assemble just in time instead)
        PROF(3);
        __threadfence_block();
        __syncthreads();
        TRC(5, trc_sum(w.Td, (size_t)d.nj * d.ldb * d.ldb, red));
        if (!factor_entry(ba, asm_args(c), lA, flag)) {
            fail_reason = 2;  // Newton matrix not positive definite
            break;
        }
        // logged flops of the factorisation: tiled path Cholesky + coupling solve + rank-k update = 7/3 nk^3 per knot; wave path L D L'
        // (1/3), the triangular inverse M = L^-T (1/3), the coupling rows (3 multiply-adds per entry) and the rank-nk update on the
        // columns where X is structurally non-zero (1/2): 7/6 nk^3 + 6 nk^2
        flops += d.nk <= 36 ? (double)d.nj * ((7.0 / 6.0) * d.nk * (double)d.nk * d.nk + 6.0 * d.nk * (double)d.nk)
                            : (double)d.nj * (7.0 / 3.0) * d.nk * (double)d.nk * d.nk;
        PROF(4);
        TRC(6, trc_sum(d.nk <= 36 ? w.Lf : w.Td, d.nk <= 36 ? (size_t)d.nj * 2 * d.nk * d.nk : (size_t)d.nj * d.ldb * d.ldb, red));
        // ---- predictor
#if QP_FUSE_F && !defined(QP_TRACE)
        if (d.nk <= 36) {
            if (QP_FUSE_F < 2) {
                rhs_from_acc(c, false, 0.0);
                __threadfence_block();
                __syncthreads();
            }
            PROF(5);
            // (QP_FUSE_F = 2: starts with rhs_from_acc into the LDS vector; ends with apply_F from it, fence and barrier)
            solve_entry(ba, w.rhs, lA, SolveOut{w.dxa, w.Lk, d.oq, QP_FUSE_F > 1 ? 1 : 0, w.cpacc, w.rbase, 0.0});
            PROF(6);
        } else
#endif
        {
            rhs_from_acc(c, false, 0.0);  // rhs = rbase + F'G'v (v from the build sweep)
            __threadfence_block();
            __syncthreads();
            PROF(5);
            TRC(7, trc_sum(w.rhs, (size_t)d.nj * d.nk, red));
            solve_entry(ba, w.rhs, lA);
            PROF(6);
            TRC(8, trc_sum(w.rhs, (size_t)d.nj * d.nk, red));
            apply_F(d, w, w.rhs, w.dxa);
            __threadfence_block();
            __syncthreads();
        }
        io.sum0 = io.sum1 = io.sum2 = 0, io.vmax = 1.0;  // vmax = max(1, max -d/x): a_aff = min(1, min -x/d)
        PROF(5);
        SWEEP(PASS_AFF);
        PROF(7);
        const double a_aff = 1.0 / block_reduce(io.vmax, 1, red);
        const double q0 = block_reduce(io.sum0, 0, red), q1 = block_reduce(io.sum1, 0, red), q2 = block_reduce(io.sum2, 0, red);
        const double mu_aff = (q0 + a_aff * q1 + a_aff * a_aff * q2) / nrows_free;
        TRC(9, a_aff); TRC(10, mu_aff);
        double sigma = mu_aff / mu;
        sigma = QP_SIGMA_POW == 2 ? sigma * sigma : (QP_SIGMA_POW == 4 ? sigma * sigma * sigma * sigma : sigma * sigma * sigma);
        io.sigma_mu = sigma * mu;
        // ---- corrector (its right-hand side was accumulated by the AFF sweep in two parts)
        __threadfence_block();
        __syncthreads();
        PROF(8);
#if QP_FUSE_F && !defined(QP_TRACE)
        if (d.nk <= 36) {
            if (QP_FUSE_F < 2) {
                rhs_from_acc(c, true, io.sigma_mu);
                __threadfence_block();
                __syncthreads();
            }
            PROF(5);
            solve_entry(ba, w.rhs, lA, SolveOut{w.dx, w.Lk, d.oq, QP_FUSE_F > 1 ? 2 : 0, w.cpacc, w.rbase, io.sigma_mu});
            PROF(6);
        } else
#endif
        {
            rhs_from_acc(c, true, io.sigma_mu);
            __threadfence_block();
            __syncthreads();
            PROF(5);
            TRC(11, trc_sum(w.rhs, (size_t)d.nj * d.nk, red));
            solve_entry(ba, w.rhs, lA);
            PROF(6);
            TRC(12, trc_sum(w.rhs, (size_t)d.nj * d.nk, red));
            apply_F(d, w, w.rhs, w.dx);
            __threadfence_block();
            __syncthreads();
        }
        flops += 2.0 * d.nj * 4.0 * d.nk * (double)d.nk;
        io.vmax = QP_STEP_FRAC;  // alpha = min(1, frac * min -x/d) = frac / max(frac, max -d/x)
        PROF(5);
        SWEEP(PASS_STEP);
        PROF(9);
        double alpha = QP_STEP_FRAC / block_reduce(io.vmax, 1, red);
        TRC(13, alpha);
        __threadfence_block();
        __syncthreads();
        // ---- step, wide neighbourhood (no product below 1e-3 * mu(alpha)) and the next iteration's first sweep in ONE pass:
        // x += alpha dx is applied speculatively, the sweep reads the OLD row state, recomputes the step, writes the NEW state
        // into the second pair of arrays, evaluates the neighbourhood test on it and builds the next iteration's weights and
        // residuals.  In the rare case the test fails the sweep is repeated from the (untouched) old state with 0.8 alpha.
        double applied = 0;
        for (int bt = 0; bt < 40; ++bt) {
            const double delta = alpha - applied;
            {  // (the batch agents' control points are contiguous: [first .. first + nb) x 3 x oq; four entries per thread and round, loads first)
                double* xb = ctrl + (size_t)first * 3 * d.oq;
                const int nx = d.nb * 3 * d.oq;
                for (int i0 = tid; i0 < nx; i0 += 4 * QP_THREADS) {
                    double xv[4], dv4[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int i = i0 + q * QP_THREADS;
                        xv[q] = i < nx ? xb[i] : 0.0, dv4[q] = i < nx ? w.dx[i] : 0.0;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int i = i0 + q * QP_THREADS;
                        if (i < nx) xb[i] = xv[q] + delta * dv4[q];
                    }
                }
            }
            __threadfence_block();
            __syncthreads();
            io.alpha = alpha, io.sum0 = 0, io.vmax = 0, io.vmin = 1e300;
            SWEEP(PASS_UPBUILD);
            applied = alpha;
            gap_next = block_reduce(io.sum0, 0, red);
            pres_next = block_reduce(io.vmax, 1, red);
            const double pmin = block_reduce(io.vmin, 2, red);
            rows_swept += nrows_free;
            __threadfence_block();
            __syncthreads();
            if (pmin >= QP_NBHD_GAMMA * gap_next / nrows_free) break;
            alpha *= QP_NBHD_BACKOFF;
        }
        // the new state becomes the current one
        {
            double* t0 = c.w.s;
            c.w.s = c.w.s2, c.w.s2 = t0;
            t0 = c.w.z, c.w.z = c.w.z2, c.w.z2 = t0;
            if (tid == 0) c_lds.w.s = c.w.s, c_lds.w.s2 = c.w.s2, c_lds.w.z = c.w.z, c_lds.w.z2 = c.w.z2;
            __syncthreads();
        }
        rows_swept += 2 * nrows_free;
        row_bytes += bytes_iter, sweep_bytes += bytes_sweeps;
        PROF(10);
        PROF(11);
    }
    if (!ok && nfar > 0) {  // the interior-point method failed on the reduced row set: once more with every row
        if (tid == 0) scal[SC_IPM_ITERS] += it_count, scal[SC_FLOPS] += flops, scal[SC_ROWS] += rows_swept;
        return 1;
    }
    if (!ok) {
        if (tid == 0) {
            atomicCAS(&S.status[mission], 0, (int)RBP_ERR_QP_FAILED);
#ifndef QP_PROFILE
            scal[SC_PROF0 + 1] = 1000.0 * batch + fail_reason, scal[SC_PROF0 + 2] = it_count;  // why: 2 factor, 3 iteration cap
#endif
        }
        return 0;
    }
    // ---- active-set polish
    if (S.p.polish && !polished) {
        int acc = polish_entry(c, pw, lds, red2, flag2, 0);
        if (acc == 3) {  // the dual active-set solve ran out of capacity: once more with the big factor in global memory
            __syncthreads();
            acc = polish_entry(c, pw, lds, red2, flag2, 2);
        }
        polished = acc == 0 ? 1 : 0;
#ifdef QP_POLSTATS
        if (tid == 0) scal[30] += 1, scal[31] += polished;
#endif
        PROF(0);
#ifndef QP_PROFILE
        if (acc != 0 && tid == 0) scal[SC_PROF0] += 1000.0 * batch + acc;  // diagnostic: which batch was not polished, and why
#endif
        __syncthreads();
    }
    __syncthreads();
    if (S.p.polish && !polished && nfar > 0) {  // refused, and rows the interior-point phase never saw exist: this point is not an answer
        if (tid == 0) scal[SC_IPM_ITERS] += it_count, scal[SC_FLOPS] += flops, scal[SC_ROWS] += rows_swept;
        return 1;
    }
    const double kkt = polished ? red[12] : kkt_ipm;
    // objective of this batch: sum x' Q_p x  (cplex.getObjValue, :164)
    double obj = 0;
    for (int it = tid; it < d.nb * 3 * M; it += QP_THREADS) {
        const int a = it / (3 * M), k = (it / M) % 3, m = it % M;
        const double sc = w.segsc[m];
        const double* xs = ctrl + ((size_t)(first + a) * 3 + k) * d.oq + 6 * m;
        double q = 0;
        for (int i = 0; i < 6; ++i)
            for (int jj = 0; jj < 6; ++jj) q += c_Qbase[6 * i + jj] * xs[i] * xs[jj];
        obj += q * sc;
    }
    obj = block_reduce(obj, 0, red);
    if (tid == 0) {
        if (reset_cost) scal[SC_TOTAL_COST] = 0;
        scal[SC_TOTAL_COST] += obj;
        scal[SC_IPM_ITERS] += it_count;
        scal[SC_QP_SOLVED] += 1;
        scal[SC_POLISHED] += polished;
        scal[SC_KKT_MAX] = fmax(scal[SC_KKT_MAX], kkt);
        scal[SC_FLOPS] += flops;
        scal[SC_ROWS] += rows_swept;
#ifndef QP_LHSTATS
        scal[SC_ROW_BYTES] += row_bytes;
        scal[SC_SWEEP_BYTES] += sweep_bytes;
#endif
    }
    PROF_FLUSH(scal);
    return 0;
}

// the batch agents' control points back at the starting point of the first pass: build_dummy (dummy_kernel) for agents [first, first + nbmax)
__device__ void restore_dummy(const DevSession& s, int mission, int first, int nbmax) {
    const int N = s.N, MS = s.M, PS = MS + 1, M = s.Mk[mission], P = M + 1, oq = 6 * M;
    const int nb = min(nbmax, N - first);
    double* ctrl = s.ctrl + (size_t)mission * N * 3 * 6 * MS;
    for (int it = threadIdx.x; it < nb * 3 * oq; it += QP_THREADS) {
        const int j6 = it % oq, k = (it / oq) % 3, qi = first + it / (3 * oq);
        const int m = j6 / 6, j = j6 % 6;
        const float* tr = s.init_traj + (size_t)mission * N * PS * 3 + (size_t)qi * P * 3;
        const int a = (j < 3) ? 0 : 1;
        ctrl[((size_t)qi * 3 + k) * oq + j6] = (1 - a) * (double)tr[3 * m + k] + a * (double)tr[3 * (m + 1) + k];
    }
}

// The REST of a mission's schedule from batch (it0, l0) on, whose first attempt on the reduced row set failed (see qp_batch_kernel): that batch
// again with every row, then the remaining batches and passes.  Sg: the kernel's copy of the session struct in LDS (the body was written for a
// struct that lives in registers: it gets a private copy).
// ONE batch QP out of line.  (round 6) The rest-of-schedule function below used to inline the body inside its loops; with the body's calls
// compiled under interprocedural register allocation, a value the compiler had hoisted out of those loops into a vector register came back
// from a FAILED factorisation (the early way out of the interior-point loop) with the callee's leftovers in some lanes, and the next
// attempt computed addresses from it (memory fault in rbase_from_acc, found with rocgdb; -mllvm -enable-ipra=0 made it disappear at -6 % of
// the bench).  A call per batch QP gives every attempt a fresh frame: nothing of one attempt's register state reaches the next.  This path
// runs for ~1 batch QP in 800; its speed does not matter.
__device__ __noinline__ int qp_batch_body_outofline(const DevSession* Sg, double* ws_base, size_t ws_stride, int mission, int batch, int nbmax,
                                                    int reset_cost, int lds_doubles, int pass_index, double far_R) {
    const DevSession S = *Sg;
    return qp_batch_body(S, uni(ws_base), ws_stride, uni(mission), uni(batch), uni(nbmax), uni(reset_cost), uni(lds_doubles), uni(pass_index), far_R);
}
__device__ __noinline__ void qp_finish_schedule(const DevSession* Sg, double* ws_base, size_t ws_stride, int mission, int passes, int biter, int nbmax,
                                               int lds_doubles, int it0, int l0) {
    const DevSession S = *Sg;
    ws_base = uni(ws_base), mission = uni(mission), passes = uni(passes), biter = uni(biter), nbmax = uni(nbmax), lds_doubles = uni(lds_doubles);
    it0 = uni(it0), l0 = uni(l0);
    for (int it = it0; it < passes; ++it)
        for (int l = (it == it0 ? l0 : 0); l < biter; ++l) {
            const bool failed = it == it0 && l == l0;
            for (int attempt = failed ? 1 : 0; attempt < 2; ++attempt) {
                if (attempt == 1) {
#ifndef QP_PROFILE
                    if (threadIdx.x == 0) S.scalars[(size_t)mission * SC_N + SC_PROF0 + 3] += 1;
#endif
                    restore_dummy(S, mission, l * nbmax, nbmax);
                    __threadfence_block();
                    __syncthreads();
                }
                const double far_R = (attempt == 0 && it == 0 && S.p.polish && S.p.far_slack > 0.0) ? S.p.far_slack : 1e300;
                const int again = uni(qp_batch_body_outofline(Sg, ws_base, ws_stride, mission, l, nbmax, (int)(l == 0), lds_doubles, it, far_R));
                __threadfence_block();
                __syncthreads();
                if (!again) break;
            }
        }
}

// One workgroup per mission runs the WHOLE Gauss-Seidel schedule of solveQP (rbp_planner.hpp:140-203: `passes` sweeps
// over `biter` batches) in one launch: missions are independent, so nothing forces them to wait for each other at batch
// boundaries (a launch per batch costs the sum over batches of the slowest mission's interior-point iteration count),
// and with more missions than CUs the hardware dispatcher balances them.
__global__ __launch_bounds__(QP_THREADS, QP_WAVES_PER_EU) void qp_batch_kernel(DevSession S, double* ws_base, size_t ws_stride, int passes,
                                                               int biter, int nbmax, int lds_doubles) {
    // longest missions of the previous run first (DevSession::qp_order): with more missions than resident workgroups the step ends when
    // the last mission does, and a long one started late leaves most of the chip idle behind it
    const int mission = S.qp_order ? __builtin_amdgcn_readfirstlane(S.qp_order[blockIdx.x]) : (int)blockIdx.x;
    const long long t_start = wall_clock64();
    // First Gauss-Seidel pass, polish on: the interior-point phase works on the reduced row set (QP_FAR_SLACK).  A batch QP that does not end
    // polished that way is solved again from its starting point -- the reference's dummy control points -- with every row: rare (1 of the 800
    // batch QPs of the 50-map sweep at 0.7 m).  The batch loop itself holds NO call: when a first attempt fails the loop stops, and the rest
    // of the mission's schedule -- that batch again with every row, then the remaining batches and passes -- runs out of line in
    // qp_finish_schedule, a stand-alone copy of the body, after the loops.  Measured, 2000 missions resident (profiles/r05_ab_reduced_rows.txt):
    // this 133.8 k agent-trajectories/s; a second attempt called from INSIDE the loop 127.7 .. 129.6 k (state kept across the call site,
    // whichever way the session struct travels), a loop of two attempts around the inlined body ~123 k, a "redo" trip through the batch loop 87 k.
    __shared__ DevSession S_lds;  // (for qp_finish_schedule only)
    if (threadIdx.x == 0) S_lds = S;
    __syncthreads();
    int stop_it = -1, stop_l = 0;
    for (int it = 0; it < passes && stop_it < 0; ++it)
        for (int l = 0; l < biter; ++l) {
            const double far_R = (it == 0 && S.p.polish && S.p.far_slack > 0.0) ? S.p.far_slack : 1e300;
            const int again = qp_batch_body(S, ws_base, ws_stride, mission, l, nbmax, (int)(l == 0), lds_doubles, it, far_R);
            __threadfence_block();
            __syncthreads();
            if (again) {
                stop_it = it, stop_l = l;
                break;
            }
        }
    if (stop_it >= 0) qp_finish_schedule(&S_lds, ws_base, ws_stride, mission, passes, biter, nbmax, lds_doubles, stop_it, stop_l);
    if (S.qp_cost && threadIdx.x == 0) S.qp_cost[mission] = (unsigned long long)(wall_clock64() - t_start) + 1;
}

// rank of every mission by the previous run's cost, longest first (ties by index); identity while any mission has no cost yet
__global__ __launch_bounds__(256) void qp_order_kernel(DevSession S) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= S.K) return;
    const unsigned long long c = S.qp_cost[k];
    int rank = 0;
    bool valid = c != 0;
    for (int o = 0; o < S.K; ++o) {
        const unsigned long long co = S.qp_cost[o];
        valid = valid && co != 0;
        rank += (co > c || (co == c && o < k)) ? 1 : 0;
    }
    // (valid is the same for every thread: each of them has looked at all costs)
    if (valid)
        S.qp_order[rank] = k;
    else
        S.qp_order[k] = k;
}

// the phase-split schedule (round 5: built, measured, loses everywhere -- DESIGN.md 3.3) is NOT part of the release library: `make dev`
// (-DRBP_PHASE_SPLIT) compiles it into lib/librbp_hip_dev.so, where rbp_solver_opts.qp_schedule = 2 selects it (tests/test_gpu_phase.py)
#if QP_THREADS == 256 && defined(RBP_PHASE_SPLIT)
#include "qp_phase.inc"
#endif

// ------------------------------------------------------------------------------------------------------------
// epilogue: Bernstein -> monomial (rbp_planner.hpp:170-196), timeScale (:209-266)
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void coef_kernel(DevSession s) {
    const int MS = s.M, oqS = 6 * MS;  // slot strides
    const size_t per_mission = (size_t)s.N * 3 * MS, total = (size_t)s.K * per_mission;
    for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (size_t)gridDim.x * blockDim.x) {
        const int mission = (int)(it / per_mission);
        if (s.status[mission] != 0) continue;
        const int M = s.Mk[mission], oq = 6 * M;
        const size_t rest = it % per_mission;
        const int m = (int)(rest % MS);
        const size_t u = rest / MS;  // agent*3 + k
        if (m >= M) continue;
        const double* T = s.T + (size_t)mission * (MS + 1);
        const double inv = 1.0 / (T[m + 1] - T[m]);
        const size_t base = (size_t)mission * s.N * 3 * oqS + u * oq + 6 * m;
        const double* v = s.ctrl + base;
        double* out = s.coef + base;
        for (int cidx = 0; cidx < 6; ++cidx) {
            const double tp = pow(inv, 5 - cidx);  // timeMatrix :695-700
            double acc = 0;
            for (int i = 0; i < 6; ++i) acc = acc + v[i] * (c_basis[6 * i + cidx] * tp);
            out[cidx] = acc;
        }
    }
}

__device__ inline int coef_derivative(int i, int j) {  // :721-723
    int r = 1;
    for (int t = 0; t < i; ++t) r *= (j - t);
    return r;
}

// real roots of c0 t^3 + c1 t^2 + c2 t + c3 after stripping leading zeros (roots_derivative :727-754).
// DEVIATION shared with the oracle: all real roots are used (the reference inspects the first two eigenvalues of
// Eigen's companion-matrix solver in Eigen's internal order, which cannot be reproduced without Eigen).
__device__ int real_roots(const double* cin, int deg, double* out) {
    const double* c = cin;
    while (deg > 0 && c[0] == 0) c++, deg--;
    if (deg == 0) return 0;
    if (deg == 1) {
        out[0] = -c[1] / c[0];
        return 1;
    }
    if (deg == 2) {
        const double D = c[1] * c[1] - 4 * c[0] * c[2];
        if (D < 0) return 0;
        const double sq = sqrt(D);
        out[0] = (-c[1] + sq) / (2 * c[0]), out[1] = (-c[1] - sq) / (2 * c[0]);
        return 2;
    }
    const double a = c[1] / c[0], b = c[2] / c[0], dd = c[3] / c[0];
    const double p = b - a * a / 3, qq = 2 * a * a * a / 27 - a * b / 3 + dd;
    const double disc = qq * qq / 4 + p * p * p / 27;
    int n = 0;
    if (disc > 0) {
        const double sq = sqrt(disc);
        out[n++] = cbrt(-qq / 2 + sq) + cbrt(-qq / 2 - sq) - a / 3;
    } else if (p == 0) {
        out[n++] = -a / 3;
    } else {
        const double r = sqrt(-p / 3);
        double arg = 3 * qq / (2 * p * r);
        arg = fmin(1.0, fmax(-1.0, arg));
        const double ph = acos(arg) / 3;
        for (int k = 0; k < 3; ++k) out[n++] = 2 * r * cos(ph - 2 * M_PI * k / 3) - a / 3;
    }
    for (int k = 0; k < n; ++k)
        for (int itn = 0; itn < 3; ++itn) {
            const double t = out[k], f = ((c[0] * t + c[1]) * t + c[2]) * t + c[3], fp = (3 * c[0] * t + 2 * c[1]) * t + c[2];
            if (fp != 0) out[k] = t - f / fp;
        }
    return n;
}


// ---- roots_derivative AS WRITTEN (rbp_planner.hpp:727-754): the real ones among the FIRST TWO eigenvalues of the companion matrix --------
// rbp_param.timescale_rule = RBP_TIMESCALE_FIRST_EIGENVALUES.  Which two of the cubic's three roots the reference's loop `j < i` (:746)
// sees is decided by the order in which Eigen::EigenSolver deflates them; Eigen is an un-vendored, un-pinned dependency of the reference,
// so its PUBLISHED algorithm (Eigen 3.3.x RealSchur: scaling by the largest |entry|, Francis double-shift QR with the deflation test
// |T(k,k-1)| <= eps (|T(k-1,k-1)| + |T(k,k)|), exceptional shifts at local iterations 10 / 30, splitOffTwoRows, eigenvalues read off the
// quasi-triangular T from the top) is restated here for matrices of order <= 3 -- a documented, deterministic order (include/rbp.h).
// The Hessenberg reduction is the identity on a companion matrix (its Householder vectors have zero tails) and is left out.
#pragma clang fp contract(off)  // the order of two nearly symmetric roots can hang on the last bit of a deflation test: no fused multiply-adds here, so that a C restatement of the same steps gives the same bits
#define ES_EPS 2.220446049250313e-16
#define ES_MIN 2.2250738585072014e-308
__device__ inline void es_householder(const double* v, int n, double* ess, double& tau, double& beta) {  // Householder.h makeHouseholder
    double tail = 0;
    for (int i = 1; i < n; ++i) tail += v[i] * v[i];
    const double c0 = v[0];
    if (tail <= ES_MIN) {
        tau = 0, beta = c0;
        for (int i = 1; i < n; ++i) ess[i - 1] = 0;
    } else {
        double b = sqrt(c0 * c0 + tail);
        if (c0 >= 0) b = -b;
        for (int i = 1; i < n; ++i) ess[i - 1] = v[i] / (c0 - b);
        tau = (b - c0) / b, beta = b;
    }
}
__device__ inline void es_house_left(double (*T)[3], int r0, int nr, int c0, int c1, const double* ess, double tau) {
    if (nr == 1) {
        for (int j = c0; j <= c1; ++j) T[r0][j] *= (1 - tau);
    } else if (tau != 0) {
        for (int j = c0; j <= c1; ++j) {
            double tmp = 0;
            for (int i = 1; i < nr; ++i) tmp += ess[i - 1] * T[r0 + i][j];
            tmp += T[r0][j];
            T[r0][j] -= tau * tmp;
            for (int i = 1; i < nr; ++i) T[r0 + i][j] -= tau * ess[i - 1] * tmp;
        }
    }
}
__device__ inline void es_house_right(double (*T)[3], int r0, int r1, int c0, int nc, const double* ess, double tau) {
    if (nc == 1) {
        for (int i = r0; i <= r1; ++i) T[i][c0] *= (1 - tau);
    } else if (tau != 0) {
        for (int i = r0; i <= r1; ++i) {
            double tmp = 0;
            for (int j = 1; j < nc; ++j) tmp += T[i][c0 + j] * ess[j - 1];
            tmp += T[i][c0];
            T[i][c0] -= tau * tmp;
            for (int j = 1; j < nc; ++j) T[i][c0 + j] -= tau * tmp * ess[j - 1];
        }
    }
}
// eigenvalues of the n x n (n <= 3) upper Hessenberg A in EigenSolver's order; false = the QR iteration did not converge
__device__ bool es_eigenvalues(const double (*A)[3], int n, double* re, double* im) {
    double T[3][3], scale = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) scale = fmax(scale, fabs(A[i][j]));
    if (scale < ES_MIN) {
        for (int i = 0; i < n; ++i) re[i] = 0, im[i] = 0;
        return true;
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) T[i][j] = A[i][j] / scale;
    double norm = 0;  // computeNormOfT
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < (j + 2 < n ? j + 2 : n); ++i) norm += fabs(T[i][j]);
    int iu = n - 1, iter = 0, total = 0;
    double exshift = 0;
    if (norm != 0)
        while (iu >= 0) {
            int il = iu;  // findSmallSubdiagEntry
            while (il > 0) {
                const double sd = fabs(T[il - 1][il - 1]) + fabs(T[il][il]);
                if (fabs(T[il][il - 1]) <= ES_EPS * sd) break;
                il--;
            }
            if (il == iu) {  // one root
                T[iu][iu] += exshift;
                if (iu > 0) T[iu][iu - 1] = 0;
                iu--, iter = 0;
            } else if (il == iu - 1) {  // splitOffTwoRows
                const double p = 0.5 * (T[iu - 1][iu - 1] - T[iu][iu]);
                const double q = p * p + T[iu][iu - 1] * T[iu - 1][iu];
                T[iu][iu] += exshift, T[iu - 1][iu - 1] += exshift;
                if (q >= 0) {
                    const double z = sqrt(fabs(q));
                    const double gp = (p >= 0) ? p + z : p - z, gq = T[iu][iu - 1];
                    double c, sn;  // JacobiRotation::makeGivens (real)
                    if (gq == 0) {
                        c = gp < 0 ? -1 : 1, sn = 0;
                    } else if (gp == 0) {
                        c = 0, sn = gq < 0 ? 1 : -1;
                    } else if (fabs(gp) > fabs(gq)) {
                        const double t = gq / gp;
                        double u = sqrt(1 + t * t);
                        if (gp < 0) u = -u;
                        c = 1 / u, sn = -t * c;
                    } else {
                        const double t = gp / gq;
                        double u = sqrt(1 + t * t);
                        if (gq < 0) u = -u;
                        sn = -1 / u, c = -t * sn;
                    }
                    for (int j = iu - 1; j < n; ++j) {  // rightCols(size - iu + 1).applyOnTheLeft(iu - 1, iu, rot.adjoint())
                        const double x = T[iu - 1][j], y = T[iu][j];
                        T[iu - 1][j] = c * x - sn * y, T[iu][j] = sn * x + c * y;
                    }
                    for (int i = 0; i <= iu; ++i) {  // topRows(iu + 1).applyOnTheRight(iu - 1, iu, rot)
                        const double x = T[i][iu - 1], y = T[i][iu];
                        T[i][iu - 1] = c * x - sn * y, T[i][iu] = sn * x + c * y;
                    }
                    T[iu][iu - 1] = 0;
                }
                if (iu > 1) T[iu - 1][iu - 2] = 0;
                iu -= 2, iter = 0;
            } else {  // il < iu - 1: only n = 3, il = 0, iu = 2
                double sh0 = T[iu][iu], sh1 = T[iu - 1][iu - 1], sh2 = T[iu][iu - 1] * T[iu - 1][iu];  // computeShift
                if (iter == 10) {
                    exshift += sh0;
                    for (int i = 0; i <= iu; ++i) T[i][i] -= sh0;
                    const double sd = fabs(T[iu][iu - 1]) + fabs(T[iu - 1][iu - 2]);
                    sh0 = 0.75 * sd, sh1 = 0.75 * sd, sh2 = -0.4375 * sd * sd;
                }
                if (iter == 30) {
                    double sd = (sh1 - sh0) / 2.0;
                    sd = sd * sd + sh2;
                    if (sd > 0) {
                        sd = sqrt(sd);
                        if (sh1 < sh0) sd = -sd;
                        sd = sd + (sh1 - sh0) / 2.0;
                        sd = sh0 - sh2 / sd;
                        exshift += sd;
                        for (int i = 0; i <= iu; ++i) T[i][i] -= sd;
                        sh0 = sh1 = sh2 = 0.964;
                    }
                }
                iter++, total++;
                if (total > 40 * n) return false;
                int imm;  // initFrancisQRStep
                double v[3] = {0, 0, 0};
                for (imm = iu - 2; imm >= il; --imm) {
                    const double Tmm = T[imm][imm], r = sh0 - Tmm, sd = sh1 - Tmm;
                    v[0] = (r * sd - sh2) / T[imm + 1][imm] + T[imm][imm + 1];
                    v[1] = T[imm + 1][imm + 1] - Tmm - r - sd;
                    v[2] = T[imm + 2][imm + 1];
                    if (imm == il) break;
                    const double lhs = T[imm][imm - 1] * (fabs(v[1]) + fabs(v[2]));
                    const double rhs = v[0] * (fabs(T[imm - 1][imm - 1]) + fabs(Tmm) + fabs(T[imm + 1][imm + 1]));
                    if (fabs(lhs) < ES_EPS * rhs) break;
                }
                for (int k = imm; k <= iu - 2; ++k) {  // performFrancisQRStep
                    const bool first = (k == imm);
                    double w[3], ess[2], tau, beta;
                    if (first)
                        w[0] = v[0], w[1] = v[1], w[2] = v[2];
                    else
                        w[0] = T[k][k - 1], w[1] = T[k + 1][k - 1], w[2] = T[k + 2][k - 1];
                    es_householder(w, 3, ess, tau, beta);
                    if (beta != 0) {
                        if (first && k > il)
                            T[k][k - 1] = -T[k][k - 1];
                        else if (!first)
                            T[k][k - 1] = beta;
                        es_house_left(T, k, 3, k, n - 1, ess, tau);
                        es_house_right(T, 0, (iu < k + 3 ? iu : k + 3), k, 3, ess, tau);
                    }
                }
                {
                    double w[2] = {T[iu - 1][iu - 2], T[iu][iu - 2]}, ess[1], tau, beta;
                    es_householder(w, 2, ess, tau, beta);
                    if (beta != 0) {
                        T[iu - 1][iu - 2] = beta;
                        es_house_left(T, iu - 1, 2, iu - 1, n - 1, ess, tau);
                        es_house_right(T, 0, iu, iu - 1, 2, ess, tau);
                    }
                }
                for (int i = imm + 2; i <= iu; ++i) {  // clean up pollution due to round-off errors
                    T[i][i - 2] = 0;
                    if (i > imm + 2) T[i][i - 3] = 0;
                }
            }
        }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) T[i][j] *= scale;
    for (int i = 0; i < n; ++i) {  // EigenSolver::compute: eigenvalues from the quasi-triangular T, top to bottom
        if (i == n - 1 || T[i + 1][i] == 0) {
            re[i] = T[i][i], im[i] = 0;
        } else {
            const double p = 0.5 * (T[i][i] - T[i + 1][i + 1]);
            double t0 = T[i + 1][i], t1 = T[i][i + 1];
            const double mx = fmax(fabs(p), fmax(fabs(t0), fabs(t1)));
            t0 /= mx, t1 /= mx;
            const double p0 = p / mx, z = mx * sqrt(fabs(p0 * p0 + t0 * t1));
            re[i] = T[i + 1][i + 1] + p, im[i] = z;
            re[i + 1] = T[i + 1][i + 1] + p, im[i + 1] = -z;
            ++i;
        }
    }
    return true;
}
// roots_derivative(i, coef_der) for the polynomial c[0] t^deg + ... + c[deg]: the real ones among the first `take` eigenvalues (the reference
// passes the derivative order i = 2 as that bound, :746; with fewer eigenvalues than that it reads past the end -- guarded: j < deg)
__device__ int first_eigen_roots(const double* cin, int deg, int take, double* out) {
    const double* c = cin;
    while (deg > 0 && c[0] == 0) c++, deg--;  // :729-733
    if (deg == 0) return 0;
    double A[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, re[3], im[3];
    for (int j = 0; j < deg; ++j) {  // :737-744
        if (j < deg - 1) A[j + 1][j] = 1;
        A[0][j] = -c[j + 1] / c[0];
    }
    if (!es_eigenvalues(A, deg, re, im)) return 0;
    int n = 0;
    for (int j = 0; j < take && j < deg; ++j)
        if (im[j] == 0) out[n++] = re[j];
    return n;
}
#pragma clang fp contract(fast)

// scale_to_max_vel :756-794 with the candidate roots of one rule
__device__ double vel_scale(const double (*cd)[6], const double* roots, int nroots, double dt, double lim) {
    const int n = 5;
    double tsx[8];
    int nt = 0;
    for (int a = 0; a < nroots; ++a) tsx[nt++] = roots[a];
    tsx[nt++] = 0, tsx[nt++] = dt;
    double vel_max = 0, t_max = 0;
    for (int a = 0; a < nt; ++a) {
        const double t = tsx[a];
        if (t < 0 || t > dt) continue;
        double vel = 0;
        for (int i = 0; i <= n - 1; ++i) vel += cd[1][i] * pow(t, n - 1 - i);
        vel = fabs(vel);
        if (vel_max < vel) vel_max = vel, t_max = t;
    }
    double sc = 1;
    while (vel_max > lim && sc < 1e6) {
        sc *= 1.1;
        double vel = 0;
        for (int i = 0; i <= n - 1; ++i) vel += cd[1][i] * pow(1 / sc, n - i) * pow(t_max, n - 1 - i);
        vel_max = fabs(vel);
    }
    return sc;
}

// one workgroup per mission: max over (agent, dim, segment) of the per-segment scale under BOTH rules of rbp_param.timescale_rule, then
// rescale by the selected one
__global__ __launch_bounds__(256) void timescale_kernel(DevSession s) {
    const int mission = blockIdx.x, tid = threadIdx.x, M = s.Mk[mission], MS = s.M, N = s.N, oq = 6 * M, n = 5;
    if (s.status[mission] != 0) return;
    __shared__ double red[8];
    const double* T = s.T + (size_t)mission * (MS + 1);
    double* coef = s.coef + (size_t)mission * N * 3 * 6 * MS;
    double ts = 1, ts1 = 1;  // all real roots / the first two eigenvalues
    if (s.p.time_scale) {
        for (int it = tid; it < N * 3 * M; it += blockDim.x) {
            const int qi = it / (3 * M), k = (it / M) % 3, m = it % M;
            const double* cf = coef + ((size_t)qi * 3 + k) * oq + 6 * m;
            double cd[4][6];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 6; ++j) cd[i][n - j] = (i <= j) ? coef_derivative(i, j) * cf[n - j] : 0.0;
            const double dt = T[m + 1] - T[m];
            // Shortcut (round 6, when both root rules became part of every run): a Bezier curve stays inside the hull of its control points, so
            // |velocity| <= 5 max |c_{i+1} - c_i| / dt and |acceleration| <= 20 max |c_{i+2} - 2 c_{i+1} + c_i| / dt^2 on the whole segment.
            // A segment whose bound is inside the limit (with a margin far above what the power-basis sums below can differ by) scales by 1
            // under either rule, whatever its roots: no root finding, no eigenvalues.  Everything else takes the reference's path unchanged.
            const double* cp = s.ctrl + (size_t)mission * N * 3 * 6 * MS + ((size_t)qi * 3 + k) * oq + 6 * m;
            double vb = 0, ab = 0;
            for (int i = 0; i < 5; ++i) vb = fmax(vb, fabs(cp[i + 1] - cp[i]));
            for (int i = 0; i < 4; ++i) ab = fmax(ab, fabs(cp[i + 2] - 2 * cp[i + 1] + cp[i]));
            const bool vel_inside = vb * (5.0 / dt) * (1 + 1e-9) <= s.max_vel[((size_t)mission * N + qi) * 3 + k];
            const bool acc_inside = ab * (20.0 / (dt * dt)) * (1 + 1e-9) <= s.max_acc[((size_t)mission * N + qi) * 3 + k];
            if (!vel_inside) {  // scale_to_max_vel :756-794, under both root rules
                double r0[3], r1[3];
                const int n0 = real_roots(cd[2], 3, r0), n1 = first_eigen_roots(cd[2], 3, 2, r1);  // roots_derivative(2, coef_der) :761
                const double lim = s.max_vel[((size_t)mission * N + qi) * 3 + k];
                const double sc0 = vel_scale(cd, r0, n0, dt, lim);
                bool same = n0 == n1;
                for (int a = 0; same && a < n0; ++a) same = r0[a] == r1[a];
                const double sc1 = same ? sc0 : vel_scale(cd, r1, n1, dt, lim);
                if (ts < sc0) ts = sc0;
                if (ts1 < sc1) ts1 = sc1;
            }
            if (!acc_inside) {  // scale_to_max_acc :797-847
                const double a = cd[3][0], b = cd[3][1], cc = cd[3][2], D = b * b - 4 * a * cc;
                double tsx[4] = {0, dt, 0, 0};
                int nt = 2;
                if (D >= 0 && a != 0) {
                    tsx[nt++] = (-b + sqrt(D)) / (2 * a);
                    tsx[nt++] = (-b - sqrt(D)) / (2 * a);
                } else if (a == 0 && b != 0)
                    tsx[nt++] = -cc / b;
                double acc_max = 0, t_max = 0;
                for (int e = 0; e < nt; ++e) {
                    const double t = tsx[e];
                    if (t < 0 || t > dt) continue;
                    double acc = 0;
                    for (int i = 0; i < 4; ++i) acc += cd[2][i] * pow(t, 3 - i);
                    acc = fabs(acc);
                    if (acc_max < acc) acc_max = acc, t_max = t;
                }
                double sc = 1;
                const double lim = s.max_acc[((size_t)mission * N + qi) * 3 + k];
                while (acc_max > lim && sc < 1e6) {
                    sc *= 1.1;
                    double acc = 0;
                    for (int i = 0; i < 4; ++i) acc += cd[2][i] * pow(1 / sc, n - i) * pow(t_max, 3 - i);
                    acc_max = fabs(acc);
                }
                if (ts < sc) ts = sc;
                if (ts1 < sc) ts1 = sc;
            }
        }
        // block max
        for (int o = 32; o > 0; o >>= 1) ts = fmax(ts, __shfl_xor(ts, o)), ts1 = fmax(ts1, __shfl_xor(ts1, o));
        if ((tid & 63) == 0) red[tid >> 6] = ts, red[4 + (tid >> 6)] = ts1;
        __syncthreads();
        ts = red[0], ts1 = red[4];
        for (int i = 1; i < (int)blockDim.x / 64; ++i) ts = fmax(ts, red[i]), ts1 = fmax(ts1, red[4 + i]);
        __syncthreads();
        if (s.p.timescale_rule == RBP_TIMESCALE_FIRST_EIGENVALUES) {
            const double t = ts;
            ts = ts1, ts1 = t;
        }
    }
    if (tid == 0) s.scalars[(size_t)mission * SC_N + SC_TIME_SCALE] = ts, s.scalars[(size_t)mission * SC_N + SC_TIME_SCALE_ALT] = ts1;
    // :236-265.  Only the coefficients are rescaled on the device: T, the SFC end times and the RSFC times stay as uploaded /
    // as the corridor stage wrote them (so a session can be re-run without restoring anything) and rbp_session_download
    // multiplies its host copies by time_scale -- the same IEEE product the reference computes in place (:250-264).
    if (ts != 1) {
        for (int it = tid; it < N * 3 * oq; it += blockDim.x) {
            const int i = it % 6;
            coef[it] = pow(1.0 / ts, n - i) * coef[it];
        }
    }
}

}  // namespace

// This file is compiled twice (csrc/Makefile), both with 256 VGPRs per lane: QP_THREADS=512 (one workgroup per CU: the fastest single
// mission, its sweeps prefetch four rows ahead) and QP_THREADS=256 (two workgroups per CU: the throughput build, used when there are
// more missions than CUs).  QP_SUFFIX names the entry points; abi/session.hip picks one per launch.
#ifndef QP_SUFFIX
#define QP_SUFFIX _w2
#endif
#define QP_CAT2(a, b) a##b
#define QP_CAT(a, b) QP_CAT2(a, b)
size_t QP_CAT(planner_workspace_bytes, QP_SUFFIX)(int N, int M, int batch_size_eff) {
    return (ws_doubles(N, M, batch_size_eff) * sizeof(double) + 255) & ~size_t(255);
}

#if QP_THREADS == 512
// build_dummy in front of, Bernstein -> monomial + timeScale behind the grid-wide joint QP (kernels/jqp.hip), which has no copies of
// these kernels (exported by the 512-thread build only)
void launch_planner_prologue(const DevSession& s, hipStream_t st) {
    const size_t total = (size_t)s.K * s.N * 3 * 6 * s.M;
    hipLaunchKernelGGL(dummy_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, st, s);
}
void launch_planner_epilogue(const DevSession& s, hipStream_t st) {
    const size_t tot2 = (size_t)s.K * s.N * 3 * s.M;
    hipLaunchKernelGGL(coef_kernel, dim3((unsigned)std::min<size_t>((tot2 + 255) / 256, 4096)), dim3(256), 0, st, s);
    hipLaunchKernelGGL(timescale_kernel, dim3(s.K), dim3(256), 0, st, s);
}
#endif

// dynamic LDS of the QP kernels: the largest of the tiled path's vectors, the polish (dual factor + 3x3 chain factor) and the wave
// path's staged substitutions (a short last batch may take the wave path even when bs > 4)
static size_t qp_lds_bytes(int bs, int M) {
    const int nk = 9 * bs;
    const int nkw = std::min(nk, 36);
    size_t lds = sizeof(double) * (2 * (size_t)((nk + 15) & ~15) + QP_THREADS + 32) + 16;
    lds = std::max(lds, sizeof(double) * (size_t)(std::max(polish_lds_doubles(nk), polish_lds_doubles(nkw)) + 18 * (M - 1) + 32) + 16);
    lds = std::max(lds, sizeof(double) * (16 + (size_t)QP_STAGE_BUFS * 2 * (nkw * KL_LD + KL_I) + (size_t)(M - 1) * nkw + 2 + 6 * KS_VLEN + 2 * 64 + 64));  // solve_staged
    // chain areas + assembly progress counters (+ the assembling waves' LDS scratch in the 512-thread build)
    // (256-thread build: the two companion waves' images of the just-in-time assembly)
    lds = std::max(lds, sizeof(double) * (size_t)(2 * kl_area_doubles(nkw) + 128 + (QP_FOLLOW_ASM ? 2 : ASM_HELPERS) * ASML_DOUBLES(nkw, nkw / 9) + 32) + 16);
    if (nk > 36 && nk <= 72) {  // LDS-resident tiled path: three blocks of a knot (leading dimension + 2)
        const size_t lb = (size_t)((nk + 15) & ~15);
        lds = std::max(lds, sizeof(double) * (3 * lb * (lb + 2) + 34) + 16);
    }
    return lds;
}

void QP_CAT(launch_planner, QP_SUFFIX)(const DevSession& s, void* qp_ws, size_t ws_bytes_per_mission, hipStream_t st) {
    const int N = s.N, M = s.M;
    // setBatch (rbp_planner.hpp:849-872)
    int bs = s.p.sequential ? s.p.batch_size : N;
    if (bs <= 0) bs = 1;
    if (bs > N) bs = N;
    const int bmax = (N + bs - 1) / bs;
    int biter = s.p.sequential ? s.p.batch_iter : 1;
    if (s.p.sequential && (biter < 0 || biter > bmax)) biter = bmax;
    const size_t total = (size_t)s.K * N * 3 * 6 * M;  // M = s.M: the slot stride (largest M of the session)
    const unsigned g = (unsigned)std::min<size_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(dummy_kernel, dim3(g), dim3(256), 0, st, s);
    if (biter > 0) {
        if (bs > QP_MAX_NB) {
            // a batch wider than QP_MAX_NB agents (nk > 576: the joint QP of a mission with more than 64 agents) is refused by
            // rbp_session_run before any launch (abi/session.hip); reaching this line is an internal error
            (void)rbp_set_error(RBP_ERR_BAD_ARGUMENT, "launch_planner: batch wider than the QP kernel supports");
            return;
        }
        const size_t lds = qp_lds_bytes(bs, M);
        (void)hipFuncSetAttribute((const void*)qp_batch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (s.p.iteration > 0 && s.qp_order) hipLaunchKernelGGL(qp_order_kernel, dim3((s.K + 255) / 256), dim3(256), 0, st, s);
        if (s.p.iteration > 0)
            hipLaunchKernelGGL(qp_batch_kernel, dim3(s.K), dim3(QP_THREADS), lds, st, s, (double*)qp_ws,
                               ws_bytes_per_mission / sizeof(double), s.p.iteration, biter, bs, (int)(lds / sizeof(double)) - 2);
    }
    const size_t tot2 = (size_t)s.K * N * 3 * M;
#ifdef QP_TRACE
    return;  // the coef slot holds the trace
#endif
    hipLaunchKernelGGL(coef_kernel, dim3((unsigned)std::min<size_t>((tot2 + 255) / 256, 4096)), dim3(256), 0, st, s);
    hipLaunchKernelGGL(timescale_kernel, dim3(s.K), dim3(256), 0, st, s);
}

#if QP_THREADS == 256 && !defined(RBP_PHASE_SPLIT)
void launch_planner_phased(const DevSession&, void*, size_t, hipStream_t, hipStream_t*, hipEvent_t*, int, int) {}  // (never reached: abi/session.hip refuses qp_schedule = 2)
bool planner_has_phase_split() { return false; }
#endif
#if QP_THREADS == 256 && defined(RBP_PHASE_SPLIT)
bool planner_has_phase_split() { return true; }
// The PHASE-SPLIT schedule (kernels/qp_phase.inc): a fixed budget of rounds is enqueued on G streams (one group of missions each, forked
// from and joined back into the caller's stream by events), nothing is synchronised.  rounds <= 0: the default budget of
// QP_ROUNDS_PER_QP rounds per batch QP of the schedule (a batch QP takes ~18 interior-point iterations = rounds).
#ifndef QP_ROUNDS_PER_QP
#define QP_ROUNDS_PER_QP 48
#endif
void launch_planner_phased(const DevSession& s, void* qp_ws, size_t ws_bytes_per_mission, hipStream_t st, hipStream_t* gs, hipEvent_t* gev, int G,
                           int rounds) {
    const int N = s.N, M = s.M, K = s.K;
    int bs = s.p.sequential ? s.p.batch_size : N;  // setBatch (rbp_planner.hpp:849-872)
    if (bs <= 0) bs = 1;
    if (bs > N) bs = N;
    const int bmax = (N + bs - 1) / bs;
    int biter = s.p.sequential ? s.p.batch_iter : 1;
    if (s.p.sequential && (biter < 0 || biter > bmax)) biter = bmax;
    const size_t total = (size_t)K * N * 3 * 6 * M;
    hipLaunchKernelGGL(dummy_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, st, s);
    if (biter > 0 && s.p.iteration > 0) {
        if (bs > QP_MAX_NB) {
            (void)rbp_set_error(RBP_ERR_BAD_ARGUMENT, "launch_planner: batch wider than the QP kernel supports");
            return;
        }
        const size_t lds = qp_lds_bytes(bs, M);
        const int lds_doubles = (int)(lds / sizeof(double)) - 2;
        (void)hipFuncSetAttribute((const void*)ph_advance, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)ph_corr, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        double* ws = (double*)qp_ws;
        const size_t stride = ws_bytes_per_mission / sizeof(double), st_off = ph_state_offset(N, M, bs);
        if (rounds <= 0) rounds = s.p.iteration * biter * QP_ROUNDS_PER_QP + 8;
        hipLaunchKernelGGL(ph_init, dim3((K + 255) / 256), dim3(256), 0, st, s, ws, stride, st_off);
        if (G < 1) G = 1;
        if (G > K) G = K;
        const bool fork = G > 1 || gs[0] != st;
        if (fork) {
            (void)hipEventRecord(gev[0], st);
            for (int g = 0; g < G; ++g) (void)hipStreamWaitEvent(gs[g], gev[0], 0);
        }
        const int nwg = ph_nwg(bs, M);
        for (int r = 0; r < rounds; ++r)
            for (int g = 0; g < G; ++g) {
                const int m0 = (int)((long long)K * g / G), kg = (int)((long long)K * (g + 1) / G) - m0;
                hipStream_t q = gs[g];
                hipLaunchKernelGGL(ph_advance, dim3(kg), dim3(QP_THREADS), lds, q, s, ws, stride, st_off, m0, s.p.iteration, biter, bs, lds_doubles);
                hipLaunchKernelGGL(ph_sweep<PASS_AFF>, dim3(nwg, kg), dim3(256), 0, q, s, ws, stride, st_off, m0, bs);
                hipLaunchKernelGGL(ph_corr, dim3(kg), dim3(QP_THREADS), lds, q, s, ws, stride, st_off, m0, bs, lds_doubles);
                hipLaunchKernelGGL(ph_sweep<PASS_STEP>, dim3(nwg, kg), dim3(256), 0, q, s, ws, stride, st_off, m0, bs);
                hipLaunchKernelGGL(ph_ctrl, dim3(kg), dim3(QP_THREADS), 0, q, s, ws, stride, st_off, m0, bs, 0);
                hipLaunchKernelGGL(ph_sweep<PASS_UPBUILD>, dim3(nwg, kg), dim3(256), 0, q, s, ws, stride, st_off, m0, bs);
                for (int rep = 0; rep < 2; ++rep) {
                    hipLaunchKernelGGL(ph_ctrl, dim3(kg), dim3(QP_THREADS), 0, q, s, ws, stride, st_off, m0, bs, 1);
                    hipLaunchKernelGGL(ph_sweep<PASS_UPBUILD>, dim3(nwg, kg), dim3(256), 0, q, s, ws, stride, st_off, m0, bs);
                }
            }
        if (fork)
            for (int g = 0; g < G; ++g) {
                (void)hipEventRecord(gev[1 + g], gs[g]);
                (void)hipStreamWaitEvent(st, gev[1 + g], 0);
            }
        hipLaunchKernelGGL(ph_finish, dim3((K + 255) / 256), dim3(256), 0, st, s, ws, stride, st_off);
    }
    const size_t tot2 = (size_t)K * N * 3 * M;
    hipLaunchKernelGGL(coef_kernel, dim3((unsigned)std::min<size_t>((tot2 + 255) / 256, 4096)), dim3(256), 0, st, s);
    hipLaunchKernelGGL(timescale_kernel, dim3(K), dim3(256), 0, st, s);
}
#endif
