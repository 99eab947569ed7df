// A group of user-mode AQL queues of our own beside HIP's streams (round 5) for the per-step launches of the AudioDec history cone.
//
// A HIP stream runs its launches strictly one after the other (hipExtAnyOrderLaunch is ignored on gfx9: profiles/anyorder_probe.hip),
// so a dependent launch starts only when its predecessor has drained: 1.3-3.9 us of gap plus the successor's whole prologue
// (arguments, tables, first operand round trip) per level, six times per decode step.  Measured (profiles/aql_probe.cpp): the gfx950
// packet processor ALSO runs the packets of ONE queue one at a time -- barrier bit or not, acquire / release fences or not -- but
// different queues run side by side.  So the object here is a GROUP of hardware queues ("lanes") with one CU mask: consecutive
// launches of the cone go to alternating lanes, launch i+1 is dispatched, runs its prologue and waits on a device word while launch
// i still computes; the kernels order themselves (write-through stores, sharded completion counters per level, a bounded wait at
// the top of the consumer).  Queues of our own are also outside HIP's pool of hardware queues (three CU-masked HIP streams are the
// ceiling there), and the host writes a whole decode's packets at once (64 bytes each, ~0.1 us) instead of
// calling hipLaunchKernel 1200 times per decode (~3 us each: most of a core per rank).
//
// Kernels come from a code object of their own (lib/oph_cone_kernels.co, the same sources compiled --cuda-device-only), loaded
// with the HSA loader: HIP does not give out the kernel descriptors of the kernels it has loaded itself.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

namespace oph {

struct AqlKernel {
    uint64_t object = 0;                 // address of the kernel descriptor
    uint32_t kernarg_size = 0, group_static = 0, private_size = 0;
};
struct AqlQueue;

// `lanes` queues (1..4) on the HSA agent that is HIP device `hip_device` (matched by PCI address), restricted to the CUs of `cu_mask`
// (hipExtStreamCreateWithCUMask's format; mask_words = 0: no mask).  nullptr + *err on any failure: the caller keeps its HIP stream.
AqlQueue* aql_create(int hip_device, const uint32_t* cu_mask, int mask_words, const char* code_object_path, int queue_packets, int lanes, std::string* err);
void aql_destroy(AqlQueue* q);
bool aql_kernel(AqlQueue* q, const char* mangled_name, AqlKernel* out, std::string* err);
// One dispatch on lane `lane` (modulo the group's lanes): grid_wgs x 1 x 1 workgroups of block_x threads, dyn_lds bytes of dynamic LDS, kernel arguments at DEVICE address
// `kernarg` (must hold k.kernarg_size bytes, 16-byte aligned, and stay untouched until the launch has run).  barrier: wait for
// every earlier packet of this queue to complete first (what a HIP stream does).  The packet is only written; aql_ring() makes
// everything written so far visible to the packet processor.
// done_signal >= 0: the launch's completion sets dependency signal `done_signal` (aql_signals) to 0.
bool aql_dispatch(AqlQueue* q, int lane, const AqlKernel& k, uint32_t grid_wgs, uint32_t block_x, uint32_t dyn_lds, const void* kernarg, bool barrier, int done_signal = -1);
// Dependencies between lanes, kept by the packet processor (no kernel waits, no CU is held while waiting): a pool of device-only
// signals; aql_signal_arm(i) marks signal i pending (host side, before the packets that use it are rung); a launch dispatched with
// done_signal = i clears it when it completes; aql_wait_signal writes a barrier-AND packet on `lane`: the lane's later packets
// start only when signal i is clear (and, the packet carrying the barrier bit, when the lane's earlier packets have completed).
bool aql_signals(AqlQueue* q, int n);
void aql_signal_arm(AqlQueue* q, int i);
bool aql_wait_signal(AqlQueue* q, int lane, int signal);
// A RUNNING kernel may clear a signal itself: an 8-byte store of 0 (system scope) to the signal's value word
// (aql_signal_value_ptr; nullptr if the runtime does not give it out).  aql_signal_clear: the same from the host.
long long* aql_signal_value_ptr(AqlQueue* q, int i);
void aql_signal_clear(AqlQueue* q, int i);
void aql_ring(AqlQueue* q);
// packets written and not yet consumed by the packet processor (0 = the ring is empty; launches may still be running)
uint64_t aql_pending(AqlQueue* q);
// Blocks until everything submitted so far has completed (a barrier packet with a completion signal).  false on time-out / queue error.
bool aql_wait_idle(AqlQueue* q, double timeout_s);
// The group's hardware queues; aql_use_lanes: from now on logical lane i of aql_dispatch is hardware queue hw[i] (n of them).
// (The compute pipes of the chip each run one queue at a time and rotate among their busy queues every ~8 us; queues are dealt to
// the 4 pipes in creation order.  A lane that shares its pipe with a queue that is busy for the whole decode starts every launch that
// much later: the caller measures which hardware queues run freely beside its busy HIP streams and uses those.)
int aql_lanes(AqlQueue* q);
void aql_use_lanes(AqlQueue* q, int n, const int* hw);
std::string aql_state(AqlQueue* q);      // diagnostics: every lane's write / read index, the first dependency signals
const char* aql_error(AqlQueue* q);      // sticky error of the queue (asynchronous queue errors land here), or ""

}  // namespace oph
