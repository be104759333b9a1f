/* rbp_rccl.h -- the exchange hook of rbp_session_shard_joint (include/rbp.h) over RCCL / xGMI, for callers that have no torch.distributed:
 * a C++ node such as the reference's swarm_traj_planner_rbp.cpp (src/swarm_traj_planner_rbp.cpp:96-116 is where a plan is made; the reference
 * itself is single-process and has nothing to replace here -- this is the second GPU's side of ONE joint QP, rbp_planner.hpp:638-684).
 *
 * swarm_simulator_amd/lib/librbp_rccl.so (swarm_simulator_amd/csrc/rccl/, links librccl; the core library librbp_hip.so does not).
 *
 *     // rank 0:  rbp_rccl_unique_id(id);  ship the RBP_RCCL_ID_BYTES bytes to rank 1 by any means (MPI, a ROS parameter, a file)
 *     rbp_rccl_pair* pair;  rbp_rccl_pair_create(&pair, device, rank, 2, id);
 *     rbp_session_shard_joint(sess, rank, 2, rbp_rccl_exchange, pair);
 *     rbp_session_run(sess, RBP_STAGE_PLANNER, stream);          // the two chains of the knot elimination trade their data over xGMI
 *     rbp_session_shard_joint(sess, 0, 1, NULL, NULL);  rbp_rccl_pair_destroy(pair);
 *
 * rbp_rccl_exchange is an rbp_exchange_fn: ncclSend + ncclRecv of `bytes` bytes with the peer in one group on the pair's own stream, which is
 * synchronised before the call returns (the session's stream was synchronised by the library before the call).  Returns 0 on success.
 * nranks == 1 makes a pair whose peer is the rank itself (send to self): the self-test of the plumbing on a one-GPU box. */
#ifndef RBP_RCCL_H
#define RBP_RCCL_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
#define RBP_RCCL_ID_BYTES 128
typedef struct rbp_rccl_pair rbp_rccl_pair;
int rbp_rccl_unique_id(void* id_out);
int rbp_rccl_pair_create(rbp_rccl_pair** out, int device, int rank, int nranks, const void* id);
int rbp_rccl_exchange(void* pair, void* send_dev, void* recv_dev, size_t bytes);
/* the same as an rbp_exchange_stream_fn (rbp_session_shard_joint_stream): ncclSend + ncclRecv in one group ENQUEUED on `stream` (a hipStream_t, the
 * one the run was given); returns at once.  rbp_rccl_abort is the matching rbp_exchange_abort_fn (ncclCommAbort: releases both ranks). */
int rbp_rccl_exchange_stream(void* pair, void* send_dev, void* recv_dev, size_t bytes, void* stream);
int rbp_rccl_abort(void* pair);
/* an exchange that has not completed after `seconds` (default 300; <= 0: wait for ever) aborts the pair's communicator (ncclCommAbort, which
 * also releases a peer blocked in the matching receive) and fails: rbp_session_run returns RBP_ERR_EXCHANGE on both ranks.  The same
 * happens on any RCCL / HIP error of an exchange; a pair that has failed once stays failed (create a new one). */
int rbp_rccl_pair_set_timeout(rbp_rccl_pair* pair, double seconds);
void rbp_rccl_pair_destroy(rbp_rccl_pair* pair);
const char* rbp_rccl_last_error(void);
#ifdef __cplusplus
}
#endif
#endif
