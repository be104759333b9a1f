// jqp.h — the grid-wide joint QP (kernels/jqp.hip): workspace layout and launcher.
#pragma once
#include "rbp_dev.h"

#define JQ_PC 16  // partner agents per sweep thread (a control point's pair rows are split into ceil(N / JQ_PC) chunks)

// offsets (in doubles) of one mission's workspace; the same for every mission of a session (sized for the session's largest M)
struct JLayout {
    int N, MS, nkpS, nblkS, njS, nch, nred;
    size_t stride, zero_doubles;
    size_t o_state, o_segsc, o_Lk, o_Dk, o_Ek, o_boxlo, o_boxhi, o_dxa, o_dx, o_rbase, o_rhs, o_wv, o_red;
    size_t o_bs[2], o_bz[2], o_ps[2], o_pz[2], o_pwgt, o_acc, o_Y, o_P, o_scr, o_inv;
    size_t o_pol;  // polish workspace (jqp_polish.inc)
    size_t o_rhsc, o_dx2, o_tb, o_tp;  // centrality corrector: the corrector's right-hand side, the trial direction, the rows' target shifts
};

struct JArgs {
    DevSession S;
    double* ws;
    JLayout L;
    int ref_step;   // which refinement step of a Newton solve the launch belongs to (kernels of step r skip missions with ST_NREF <= r)
    int retry_only; // launches that repeat a refused step: only missions with ST_RETRY set take part
    int ref_gate;   // substitution launches of a refinement pass: missions with ST_NREF < ref_gate are skipped
    int trace;      // RBP_JOINT_TRACE: device-side diagnostics of the polish's acceptance test
    int gond_only;  // launches of the centrality corrector: only missions with ST_GACT set take part
    int chain0;     // first chain of a launch over the twisted elimination's chains (chain = blockIdx.y + chain0): 0, or the rank's own chain
                    // when the factorisation is sharded over two ranks (JointShard)
    int fuse_panel; // look-ahead schedule: jq_update forms the panel rows it needs itself (no jq_panel launch before it); set per launch
    int sweep2;     // tile sweep of the knots (kind 0): 1 = two pivot tiles per pass (jq_pivot2 / jq_panel2 / jq_update2_bulk), 0 = one
    int dreg_mode;  // 0: constant dual regularisation 1e-9 (qp.hip); 1: proximal, dreg = clamp(scale * mu, 1e-9, max)
    double dreg_scale, dreg_max;
    double gond[3];  // centrality corrector: extra step length asked for, share of it that must be gained, step length below which it is tried
    double early_mu[2];  // complementarity below which the first / the second early polish attempt is made
    double pol_tau;  // polish: a pivot of S_AA below pol_tau x the row's own diagonal marks a row that depends on the rows before it (deleted from that solve)
    double pol_vtol;  // experiment: row violation accepted after the last refinement round
    int pol_adtau;   // experiment: adaptive deletion threshold on/off
    double exit_mu;  // third exit of the interior-point loop: pres < 1e-9, dres < 1e-7, mu < exit_mu
    double pol_lh_early, pol_lh_final;  // block Lawson-Hanson rounds allowed in an early / the final polish attempt
    double tune[5];  // mu0, slack floor, centring exponent, neighbourhood gamma, step fraction
};

struct JointStats {
    int rounds;         // interior-point rounds enqueued (= host synchronisations)
    int polish_rounds;  // active-set rounds of the polish
};

// A joint solve whose twisted knot elimination is spread over TWO ranks (BASELINE config 4; rbp_session_shard_joint of include/rbp.h):
// rank r eliminates chain r and substitutes along it; everything else (row sweeps, control, polish, the middle knot) is replicated
// and bit-identical on both ranks.  Three exchanges move data between the ranks (jq_xfer packs / unpacks per mission):
// the explicit inverse of each chain's last knot before the middle knot is assembled, the forward vector of that knot before the middle
// solve, and the chain's half of the solution after the backward pass.
struct JointShard {
    int rank = 0, nranks = 1;
    double *send = nullptr, *recv = nullptr;  // device buffers of `cap` bytes each (the session's)
    size_t cap = 0;
    int (*exchange)(void* user, void* send_dev, void* recv_dev, size_t bytes) = nullptr;  // returns 0 when the peer's bytes are in recv
    // STREAM-ORDERED form (rbp_session_shard_joint_stream): only ENQUEUES the exchange on `stream` and returns; the library then neither
    // synchronises before nor after a call -- the header of every exchange is written and checked by kernels, a mismatch lands in *xerr
    // (device) and is read with the once-per-round poll of the missions' states
    int (*exchange_stream)(void* user, void* send_dev, void* recv_dev, size_t bytes, void* stream) = nullptr;
    int (*abort_peer)(void* user) = nullptr;  // optional: brings the exchange down (ncclCommAbort) when the per-round wait times out
    double timeout_s = 300.0;                 // of one round's wait (stream-ordered form)
    double* xerr = nullptr;                   // [2] device: {which header word differed (1..6; 0 = none), sequence number of that exchange}
    void* user = nullptr;
};
size_t joint_exchange_bytes(int N, int MS, int K);  // capacity the send / recv buffers of a K-mission session need

struct JointOpts {  // what rbp_solver_opts says about the grid-wide joint solver
    int corrector = 1;  // one centrality corrector per interior-point iteration
    int schedule = 0;   // tile sweep: 0 automatic, 1 look-ahead, 2 bulk, 3 bulk with two pivot tiles per pass (opt-in)
    const JointShard* shard = nullptr;  // two-rank factorisation (nullptr or nranks == 1: this rank runs both chains)
};

JLayout jq_layout(int N, int MS);
size_t joint_workspace_bytes(int N, int MS);
// dummy_kernel .. timescale_kernel around it are launched by the caller (launch_planner_prologue / _epilogue in qp.hip)
int launch_planner_joint(const DevSession& s, void* ws, hipStream_t st, JointStats* stats, const JointOpts& opts);
void launch_planner_prologue(const DevSession& s, hipStream_t st);
void launch_planner_epilogue(const DevSession& s, hipStream_t st);
