// pdeip_persist_host.hpp -- host side of the one-launch exact-order walkers (k_sor_exact_persist, k_pde8_exact_persist): the
// schedule table, the control block and the mailbox of a call.
//
// Control block (ws[WS_CTL], one memset per call from word 1 on):
//   word 0        abort (sticky: cleared only by pdeip_persist_error(), so a timed-out wait cannot be lost under the next call's reset)
//   words 4..11   one ticket counter per XCD list
//   words 16..    progress counters [nframes][iter][B]
//   128-byte aligned behind them: the west-edge mailbox
// Schedule table (ws[WS_ORDER], cached by shape): ints 0..8 = offsets of eight lists into the items, items from int 16 on, an item
// = b | (t << 16).  Every list is sorted by key = b + 2t, in which every dependency of an item -- (b-1,t), (b,t-1), (b+1,t-1) --
// has a smaller key.
//   XCD-affine (opt-in, PDEIP_PERSIST_XCD=1, and only when grid <= compute units): list x holds the strips b = x (mod 8),
//   all their sweeps; a workgroup takes the next item of the list of the XCD it runs on (HW_REG_XCC_ID) and only steals
//   from the other lists when its own is used up.  The sweeps of a strip then follow each other through ONE L2: sweep t+1 reads
//   the packed coefficients sweep t fetched 10-15 us earlier.  With every workgroup resident and exactly one item per
//   workgroup every item is taken by a running workgroup, whatever the placement: placement changes speed only.
//   Default (and always with more workgroups than compute units): one list in key order -- a running workgroup then only ever
//   waits for items with smaller tickets, which are running or finished: live for any grid, any dispatch order and whatever
//   other kernels hold compute units.
#pragma once
#include "pdeip_ctx.hpp"
#include "pdeip_sor_exact.hpp"

namespace pdeip {

constexpr int PERSIST_HDR_WORDS = 16, PERSIST_TABLE_HDR = 16;

// Schedule table on the device.  Thread (b, t) writes its item at its rank in its list: the number of items of the list that
// come before it in (key, t) order, key = b + 2t.  Items (b', t') of list x: 0 <= b' < B, b' = x (mod 8) when affine.
static __device__ __forceinline__ int persist_count_below(int lim, int B, int affine, int x)
{ // #{b' in [0, min(lim, B)) : affine ? b' % 8 == x : true}
    int n = lim < B ? lim : B;
    if (n <= 0) return 0;
    return affine ? (n - x + 7) / 8 : n;
}
// Clears the control block and the mailbox of a call (words 1 .. n-1; word 0 is the sticky abort word).  A kernel rather than
// hipMemsetAsync: an exact-order call captured into a HIP graph replayed with stale counters when the clear was a memset node.
static __global__ void k_persist_clear(unsigned *words, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (i < n) words[i] = 0u;
}
static __global__ void k_persist_order(int *table, int B, int T, int affine)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < PERSIST_TABLE_HDR) { // list offsets: list x starts behind the items of the lists before it
        int off = 0;
        for (int x = 0; x < idx && x < 8; x++) off += (affine ? (B - x + 7) / 8 : (x == 0 ? B : 0)) * T;
        table[idx] = idx <= 8 ? (idx == 8 ? B * T : off) : 0;
    }
    if (idx >= B * T) return;
    const int b = idx % B, t = idx / B, key = b + 2 * t, x = affine ? (b & 7) : 0;
    int rank = 0;
    for (int tt = 0; tt < T; tt++) {
        // items of sweep tt with a smaller key, or the same key and a smaller sweep: b' < key - 2 tt (+1 when tt < t)
        rank += persist_count_below(key - 2 * tt + (tt < t ? 1 : 0), B, affine, x);
    }
    int first = 0;
    for (int xx = 0; xx < x; xx++) first += ((B - xx + 7) / 8) * T;
    table[PERSIST_TABLE_HDR + first + rank] = b | (t << 16);
}

inline int persist_prepare(hipStream_t s, int B, int iter, int nframes, size_t mail_bytes, PersistCtl *ctl)
{
    const size_t nprog = (size_t)nframes * iter * B;
    const size_t ctl_bytes = ((PERSIST_HDR_WORDS + nprog) * sizeof(unsigned) + 127) / 128 * 128;
    float *ctl_f = nullptr, *order_f = nullptr;
    RC(ws_get(WS_CTL, ctl_bytes + mail_bytes, &ctl_f));
    RC(ws_get(WS_ORDER, (PERSIST_TABLE_HDR + (size_t)B * iter) * sizeof(int), &order_f));
    DeviceState *dst = cur_dev(); // after the ws_get calls: a regrown WS_ORDER has dropped its cached shape
    if (dst->num_cus == 0) {
        hipDeviceProp_t prop;
        dst->num_cus = (hipGetDeviceProperties(&prop, dst->device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 1;
    }
    // Default: the single key-ordered list -- a running workgroup only ever waits for smaller tickets, which are running or
    // finished, whatever else holds compute units.  The XCD-affine lists (3-5 % faster at 4K) are live only while every
    // workgroup of the grid is resident, which a library inside somebody else's process cannot know: opt-in, PDEIP_PERSIST_XCD=1.
    const int affine = (env_int("PDEIP_PERSIST_XCD", 0) != 0 && nprog <= (size_t)dst->num_cus) ? 1 : 0;
    // While the stream is being captured into a HIP graph the table is rebuilt by every call: what the cache says at capture time
    // need not be what the buffer holds when the graph is replayed (other calls may have run in between)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    if (capturing) dst->order_B = dst->order_T = 0;
    if (dst->order_B != B || dst->order_T != iter || dst->order_affine != affine) {
        // built on the call's stream by a kernel: no host table, no synchronisation, capturable into a HIP graph
        const int nitems = B * iter;
        hipLaunchKernelGGL(k_persist_order, dim3((unsigned)((nitems + 255) / 256)), dim3(256), 0, s, reinterpret_cast<int *>(order_f), B, iter, affine);
        HIPCHK(hipGetLastError());
        dst->order_B = capturing ? 0 : B; // a captured build says nothing about the buffer's content outside the graph
        dst->order_T = capturing ? 0 : iter;
        dst->order_affine = affine;
    }
    {
        const size_t nwords = (ctl_bytes + mail_bytes) / sizeof(unsigned);
        hipLaunchKernelGGL(k_persist_clear, dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0, s, reinterpret_cast<unsigned *>(ctl_f), nwords);
        HIPCHK(hipGetLastError());
    }
    ctl->abort_flag = reinterpret_cast<unsigned *>(ctl_f);
    ctl->ticket = ctl->abort_flag + 4;
    ctl->progress = ctl->abort_flag + PERSIST_HDR_WORDS;
    ctl->order = reinterpret_cast<const int *>(order_f);
    ctl->mail = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(ctl_f) + ctl_bytes);
    dst->persist_used = true;
    return PDEIP_OK;
}

} // namespace pdeip
