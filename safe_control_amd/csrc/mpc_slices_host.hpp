// Host side of the continuation launches (mpc_cont.hpp; include/safe_control_amd.h: sc_mpc_slices): carves the caller's workspace
// into [solver state B x stride | two queues of 2 x B indices | the launches' counters] and issues the launches of one call on
// the caller's stream -- no host synchronisation: every launch has a block per problem of the batch and a block whose queue slot
// is empty leaves at once.
#pragma once
#include <hip/hip_runtime.h>

#include "mpc_cont.hpp"
#include "../../include/safe_control_amd.h"

namespace sc {

inline size_t slices_align(size_t b) { return (b + 255) & ~(size_t)255; }
constexpr int SLICES_MAX_LAUNCHES = SC_MPC_MAX_SLICES + 2;             // classify + caps + the last one

inline size_t slices_workspace_bytes(long long B, size_t stride_doubles) {
    return slices_align((size_t)B * stride_doubles * sizeof(double)) + 2 * slices_align((size_t)2 * B * sizeof(int)) +
           slices_align((size_t)2 * SLICES_MAX_LAUNCHES * sizeof(int));
}

// false: a schedule the entry points reject
inline bool slices_valid(const sc_mpc_slices* sl) {
    if (!sl) return true;
    if (sl->n_caps < 0 || sl->n_caps > SC_MPC_MAX_SLICES) return false;
    for (int k = 0; k < sl->n_caps; ++k)
        if (sl->it_stop[k] < 1 || (k > 0 && sl->it_stop[k] <= sl->it_stop[k - 1])) return false;
    return (sl->order == 0 || sl->order == 1) && (sl->classify_first == 0 || sl->classify_first == 1);
}
inline bool slices_active(const sc_mpc_slices* sl, int max_iter) {
    return sl && (sl->classify_first || (sl->n_caps > 0 && sl->it_stop[0] < max_iter));
}

// launch(const ipm::Cont&) -> hipError_t issues one launch of the family's kernel over B blocks
template <typename Launch>
hipError_t run_slices(const sc_mpc_slices* sl, int max_iter, long long B, size_t stride_doubles, hipStream_t stream, Launch launch) {
    ipm::Cont ct{};
    ct.it_stop = max_iter;
    if (!slices_active(sl, max_iter) || B == 0) return launch(ct);
    char* w = (char*)sl->workspace;
    double* state = (double*)w; w += slices_align((size_t)B * stride_doubles * sizeof(double));
    int* queues[2];
    queues[0] = (int*)w; w += slices_align((size_t)2 * B * sizeof(int));
    queues[1] = (int*)w; w += slices_align((size_t)2 * B * sizeof(int));
    int* counts = (int*)w;
    hipError_t e = hipMemsetAsync(counts, 0, (size_t)2 * SLICES_MAX_LAUNCHES * sizeof(int), stream);
    if (e != hipSuccess) return e;
    int k = 0;                                                           // launches issued so far
    const int* q_in = nullptr;
    const int* c_in = nullptr;
    auto next = [&](int it_stop, int resume, bool last) {
        ct = ipm::Cont{};
        ct.state = state; ct.stride = (long long)stride_doubles;
        ct.queue_in = q_in; ct.count_in = c_in;
        ct.queue_out = queues[k & 1]; ct.count_out = counts + 2 * k; ct.qcap = B;
        ct.it_stop = it_stop; ct.resume = resume; ct.order = sl->order;
        const hipError_t el = launch(ct);
        if (!last) { q_in = ct.queue_out; c_in = ct.count_out; }
        ++k;
        return el;
    };
    int resume = 0;
    if (sl->classify_first) {
        e = next(-1, 0, false);
        if (e != hipSuccess) return e;
    }
    for (int i = 0; i < sl->n_caps && sl->it_stop[i] < max_iter; ++i) {
        e = next(sl->it_stop[i], resume, false);
        if (e != hipSuccess) return e;
        resume = 1;
    }
    return next(max_iter, resume, true);
}

}  // namespace sc
