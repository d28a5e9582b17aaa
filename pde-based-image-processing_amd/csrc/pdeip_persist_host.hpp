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
//   XCD-affine (all workgroups of the call resident at once: grid <= compute units): list x holds the strips b = x (mod 8),
//   all their sweeps; a workgroup takes the next item of the list of the XCD it runs on (HW_REG_XCC_ID) and only steals
//   from the other lists when its own is used up.  The sweeps of a strip then follow each other through ONE L2: sweep t+1 reads
//   the packed coefficients sweep t fetched 10-15 us earlier.  With every workgroup resident and exactly one item per
//   workgroup every item is taken by a running workgroup, whatever the placement: placement changes speed only.
//   Otherwise (more workgroups than compute units): one list in key order, as before -- a running workgroup then only ever
//   waits for items with smaller tickets, which are running or finished.
#pragma once
#include <vector>

#include "pdeip_ctx.hpp"
#include "pdeip_sor_exact.hpp"

namespace pdeip {

constexpr int PERSIST_HDR_WORDS = 16, PERSIST_TABLE_HDR = 16;

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
    const int affine = (env_int("PDEIP_PERSIST_XCD", 1) != 0 && nprog <= (size_t)dst->num_cus) ? 1 : 0;
    if (dst->order_B != B || dst->order_T != iter || dst->order_affine != affine) {
        std::vector<int> table(PERSIST_TABLE_HDR, 0), items;
        items.reserve((size_t)B * iter);
        for (int x = 0; x < 8; x++) {
            table[x] = (int)items.size();
            if (!affine && x > 0) continue;
            for (int key = 0; key <= (B - 1) + 2 * (iter - 1); key++)
                for (int t = 0; t < iter; t++) {
                    const int b = key - 2 * t;
                    if (b >= 0 && b < B && (!affine || (b & 7) == x)) items.push_back(b | (t << 16));
                }
        }
        table[8] = (int)items.size();
        table.insert(table.end(), items.begin(), items.end());
        HIPCHK(hipMemcpyAsync(order_f, table.data(), table.size() * sizeof(int), hipMemcpyHostToDevice, s));
        HIPCHK(hipStreamSynchronize(s)); // `table` is about to go out of scope
        dst->order_B = B;
        dst->order_T = iter;
        dst->order_affine = affine;
    }
    HIPCHK(hipMemsetAsync(reinterpret_cast<unsigned *>(ctl_f) + 1, 0, ctl_bytes - sizeof(unsigned) + mail_bytes, s));
    ctl->abort_flag = reinterpret_cast<unsigned *>(ctl_f);
    ctl->ticket = ctl->abort_flag + 4;
    ctl->progress = ctl->abort_flag + PERSIST_HDR_WORDS;
    ctl->order = reinterpret_cast<const int *>(order_f);
    ctl->mail = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(ctl_f) + ctl_bytes);
    dst->persist_used = true;
    return PDEIP_OK;
}

} // namespace pdeip
