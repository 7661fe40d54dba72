// Continuation launches of the one-NLP-per-wavefront interior-point kernels (include/safe_control_amd.h: sc_mpc_slices).
//
// A launch may stop a solve at an iteration cap: the block writes the solver state of its problem -- everything the next iteration
// reads that is not recomputed from it: the iterate, slacks, multipliers, elastic variables, best iterate / z_R, the obstacle rows
// with their barrier scales and the scalars of the loop (barrier parameter, merit penalty, inertia memory, restoration flags,
// counters) -- to `state[prob * stride ..]` in device memory, marks the problem SC_STATUS_PENDING and appends its index to the
// launch's output queue.  The next launch takes its problems from that queue, loads the state and continues with the SAME
// instructions an uninterrupted solve would have executed: resumed and uninterrupted solves agree bit for bit
// (tests/test_mpc_slices_gpu.py).  The queue has two classes; with `order` set, class 0 takes the problems whose CBF rows are
// violated at the hand-over (the ones that go on to crawl or to restore feasibility: the long solves) and the next launch starts
// them first.  A "classify only" launch (it_stop < 0) evaluates the initial guess, sorts the problems into the two classes and
// leaves; the launch after it then starts every solve from scratch, in that order -- a cheap longest-first order for the launch
// whose length is the length of its slowest problem.
#pragma once
#include <hip/hip_runtime.h>

namespace sc {
namespace ipm {

#define SC_STATUS_PENDING_MPC (-1)

struct Cont {
    double* state;          // [B][stride] doubles; nullptr: no continuation (one launch to max_iter)
    long long stride;       // doubles per problem
    const int* queue_in;    // [2][qcap] problem indices of the previous launch, nullptr: problem = block index
    const int* count_in;    // [2]
    int* queue_out;         // [2][qcap]
    int* count_out;         // [2]
    long long qcap;         // capacity of one class of a queue (= B)
    int it_stop;            // leave the solve as pending before iteration it_stop + 1 (>= max_iter: never); < 0: classify only
    int resume;             // 1: `state` holds the solver state of every problem of queue_in
    int order;              // 1: violated-at-hand-over problems into class 0, the others into class 1; 0: everything into class 0
    int reserved;
};

// the problem of this block; false: nothing to do
__device__ __forceinline__ bool cont_problem(const Cont& ct, long long B, long long& prob) {
    prob = blockIdx.x;
    if (ct.queue_in) {
        const long long c0 = ct.count_in[0], c1 = ct.count_in[1];
        if (prob >= c0 + c1) return false;
        prob = prob < c0 ? ct.queue_in[prob] : ct.queue_in[ct.qcap + (prob - c0)];
        return true;
    }
    return prob < B;
}
// one thread of the block
__device__ __forceinline__ void cont_push(const Cont& ct, long long prob, bool violated) {
    const int cls = (ct.order && !violated) ? 1 : 0;
    const int slot = atomicAdd(ct.count_out + cls, 1);
    ct.queue_out[cls * ct.qcap + slot] = (int)prob;
}
__device__ __forceinline__ void cont_copy(double* __restrict__ dst, const double* __restrict__ src, int n, int tid, int nthreads) {
    for (int i = tid; i < n; i += nthreads) dst[i] = src[i];
}
constexpr int CONT_SCALARS = 16;   // doubles at the head of a problem's state: the scalars of the loop

}  // namespace ipm
}  // namespace sc
