// oph_aql: see oph_aql.h.  HSA runtime (ROCr) calls only; HIP is used to find the PCI address of the device.
#include "oph_aql.h"

#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace oph {

struct AqlQueue {
    hsa_agent_t agent{};
    static constexpr int MAXQ = 4;
    hsa_queue_t* q[MAXQ] = {nullptr, nullptr, nullptr, nullptr};      // hardware queues ("lanes"): each runs its packets one after the other,
    int nq = 0;                                                        // different lanes run side by side
    hsa_executable_t exe{};
    hsa_code_object_reader_t reader{};
    bool have_exe = false, have_reader = false;
    hsa_signal_t idle_sig[MAXQ]{};
    bool have_sig[MAXQ] = {false, false, false, false};
    uint64_t written[MAXQ] = {0, 0, 0, 0};       // next packet index to write (== the lane's write index as we keep it)
    uint64_t rung[MAXQ] = {0, 0, 0, 0};          // doorbell value last stored + 1
    int map[MAXQ] = {0, 1, 2, 3}; int nmap = 0;  // logical lane -> hardware queue (aql_use_lanes); nmap = 0: identity over nq
    std::vector<hsa_signal_t> dep;               // dependency signals between lanes (aql_signals): 1 = pending, 0 = the producing launch has completed
    std::string error;
    std::mutex mu;
};

namespace {
std::mutex g_hsa_mu;
int g_hsa_refs = 0;

const char* st_name(hsa_status_t st) {
    const char* s = nullptr;
    return hsa_status_string(st, &s) == HSA_STATUS_SUCCESS && s ? s : "unknown HSA status";
}

struct AgentSearch { uint32_t bdf; uint32_t domain; hsa_agent_t found; bool ok; };
hsa_status_t agent_cb(hsa_agent_t a, void* data) {
    AgentSearch* s = (AgentSearch*)data;
    hsa_device_type_t type;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &type) != HSA_STATUS_SUCCESS || type != HSA_DEVICE_TYPE_GPU) return HSA_STATUS_SUCCESS;
    uint32_t bdf = 0, dom = 0;
    if (hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    (void)hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &dom);
    if (bdf == s->bdf && dom == s->domain) { s->found = a; s->ok = true; return HSA_STATUS_INFO_BREAK; }
    return HSA_STATUS_SUCCESS;
}

void queue_error_cb(hsa_status_t st, hsa_queue_t*, void* data) {
    AqlQueue* q = (AqlQueue*)data;
    if (q) {
        std::lock_guard<std::mutex> lock(q->mu);
        q->error = std::string("AQL queue error: ") + st_name(st);
    }
}
}  // namespace

AqlQueue* aql_create(int hip_device, const uint32_t* cu_mask, int mask_words, const char* code_object_path, int queue_packets, int lanes, std::string* err) {
    auto fail = [&](const std::string& m, AqlQueue* q) -> AqlQueue* { if (err) *err = m; if (q) aql_destroy(q); return nullptr; };
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, hip_device) != hipSuccess) { (void)hipGetLastError(); return fail("hipGetDeviceProperties failed", nullptr); }
    {
        std::lock_guard<std::mutex> lock(g_hsa_mu);
        const hsa_status_t st = hsa_init();        // reference-counted: HIP holds its own reference
        if (st != HSA_STATUS_SUCCESS) return fail(std::string("hsa_init: ") + st_name(st), nullptr);
        ++g_hsa_refs;
    }
    AqlQueue* q = new AqlQueue();
    AgentSearch s{(uint32_t)((prop.pciBusID << 8) | (prop.pciDeviceID << 3)), (uint32_t)prop.pciDomainID, {}, false};
    hsa_status_t st = hsa_iterate_agents(agent_cb, &s);
    if ((st != HSA_STATUS_SUCCESS && st != HSA_STATUS_INFO_BREAK) || !s.ok) return fail("no HSA agent at the HIP device's PCI address", q);
    q->agent = s.found;
    uint32_t qmax = 0;
    (void)hsa_agent_get_info(q->agent, HSA_AGENT_INFO_QUEUE_MAX_SIZE, &qmax);
    uint32_t size = 64;
    while ((int)size < queue_packets) size <<= 1;
    if (qmax && size > qmax) size = qmax;
    q->nq = lanes < 1 ? 1 : (lanes > AqlQueue::MAXQ ? AqlQueue::MAXQ : lanes);
    for (int i = 0; i < q->nq; ++i) {
        st = hsa_queue_create(q->agent, size, HSA_QUEUE_TYPE_SINGLE, queue_error_cb, q, 0, 0, &q->q[i]);
        if (st != HSA_STATUS_SUCCESS) { q->q[i] = nullptr; return fail(std::string("hsa_queue_create: ") + st_name(st), q); }
        if (mask_words > 0 && cu_mask) {
            st = hsa_amd_queue_cu_set_mask(q->q[i], (uint32_t)mask_words * 32u, cu_mask);
            if (st != HSA_STATUS_SUCCESS) return fail(std::string("hsa_amd_queue_cu_set_mask: ") + st_name(st), q);
        }
        st = hsa_signal_create(1, 0, nullptr, &q->idle_sig[i]);
        if (st != HSA_STATUS_SUCCESS) return fail(std::string("hsa_signal_create: ") + st_name(st), q);
        q->have_sig[i] = true;
    }
    // the code object
    FILE* f = fopen(code_object_path, "rb");
    if (!f) return fail(std::string("cannot open ") + code_object_path, q);
    std::vector<char> blob;
    {
        fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
        blob.resize(n > 0 ? (size_t)n : 0);
        const size_t got = blob.empty() ? 0 : fread(blob.data(), 1, blob.size(), f);
        fclose(f);
        if (got != blob.size() || blob.empty()) return fail(std::string("cannot read ") + code_object_path, q);
    }
    st = hsa_code_object_reader_create_from_memory(blob.data(), blob.size(), &q->reader);
    if (st != HSA_STATUS_SUCCESS) return fail(std::string("hsa_code_object_reader_create_from_memory: ") + st_name(st), q);
    q->have_reader = true;
    st = hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &q->exe);
    if (st != HSA_STATUS_SUCCESS) return fail(std::string("hsa_executable_create_alt: ") + st_name(st), q);
    q->have_exe = true;
    st = hsa_executable_load_agent_code_object(q->exe, q->agent, q->reader, nullptr, nullptr);
    if (st != HSA_STATUS_SUCCESS) return fail(std::string("hsa_executable_load_agent_code_object: ") + st_name(st), q);
    st = hsa_executable_freeze(q->exe, nullptr);
    if (st != HSA_STATUS_SUCCESS) return fail(std::string("hsa_executable_freeze: ") + st_name(st), q);
    for (int i = 0; i < q->nq; ++i) { q->written[i] = hsa_queue_load_write_index_relaxed(q->q[i]); q->rung[i] = q->written[i]; }
    return q;
}

void aql_destroy(AqlQueue* q) {
    if (!q) return;
    for (int i = 0; i < AqlQueue::MAXQ; ++i) if (q->q[i]) (void)hsa_queue_destroy(q->q[i]);
    if (q->have_exe) (void)hsa_executable_destroy(q->exe);
    if (q->have_reader) (void)hsa_code_object_reader_destroy(q->reader);
    for (int i = 0; i < AqlQueue::MAXQ; ++i) if (q->have_sig[i]) (void)hsa_signal_destroy(q->idle_sig[i]);
    for (hsa_signal_t s : q->dep) (void)hsa_signal_destroy(s);
    delete q;
    std::lock_guard<std::mutex> lock(g_hsa_mu);
    if (g_hsa_refs > 0) { --g_hsa_refs; (void)hsa_shut_down(); }
}

bool aql_kernel(AqlQueue* q, const char* mangled_name, AqlKernel* out, std::string* err) {
    const std::string sym = std::string(mangled_name) + ".kd";
    hsa_executable_symbol_t s{};
    hsa_status_t st = hsa_executable_get_symbol_by_name(q->exe, sym.c_str(), &q->agent, &s);
    if (st != HSA_STATUS_SUCCESS) { if (err) *err = "kernel symbol " + sym + ": " + st_name(st); return false; }
    AqlKernel k;
    if (hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object) != HSA_STATUS_SUCCESS ||
        hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg_size) != HSA_STATUS_SUCCESS ||
        hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group_static) != HSA_STATUS_SUCCESS ||
        hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.private_size) != HSA_STATUS_SUCCESS || k.object == 0) {
        if (err) *err = "kernel symbol " + sym + ": no descriptor info";
        return false;
    }
    *out = k;
    return true;
}

namespace {
// room for one more packet in the lane's ring: the packet processor advances the read index as it consumes packets
bool ring_room(AqlQueue* q, int hw_lane) {
    hsa_queue_t* hq = q->q[hw_lane];
    const auto t0 = std::chrono::steady_clock::now();
    while (q->written[hw_lane] - hsa_queue_load_read_index_scacquire(hq) >= hq->size) {
        aql_ring(q);
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 10.0) {
            std::lock_guard<std::mutex> lock(q->mu);
            if (q->error.empty()) q->error = "AQL ring stayed full for 10 s";
            return false;
        }
        std::this_thread::yield();
    }
    return true;
}
}  // namespace

bool aql_signals(AqlQueue* q, int n) {
    while ((int)q->dep.size() < n) {
        hsa_signal_t s{};
        // device-only signals: written by the packet processor at a launch's completion, read by the packet processor of another
        // lane (barrier-AND packet) -- no interrupt, no host wake-up per completion
        hsa_status_t st = hsa_amd_signal_create(1, 0, nullptr, HSA_AMD_SIGNAL_AMD_GPU_ONLY, &s);
        if (st != HSA_STATUS_SUCCESS) st = hsa_signal_create(1, 0, nullptr, &s);
        if (st != HSA_STATUS_SUCCESS) {
            std::lock_guard<std::mutex> lock(q->mu);
            if (q->error.empty()) q->error = std::string("hsa_signal_create: ") + st_name(st);
            return false;
        }
        q->dep.push_back(s);
    }
    return true;
}
void aql_signal_arm(AqlQueue* q, int i) { if (i >= 0 && i < (int)q->dep.size()) hsa_signal_store_relaxed(q->dep[i], 1); }
void aql_signal_clear(AqlQueue* q, int i) { if (i >= 0 && i < (int)q->dep.size()) hsa_signal_store_screlease(q->dep[i], 0); }
long long* aql_signal_value_ptr(AqlQueue* q, int i) {
    if (i < 0 || i >= (int)q->dep.size()) return nullptr;
    volatile hsa_signal_value_t* p = nullptr;
    if (hsa_amd_signal_value_pointer(q->dep[i], &p) != HSA_STATUS_SUCCESS) return nullptr;
    return (long long*)p;
}

bool aql_wait_signal(AqlQueue* q, int lane, int signal) {
    if (signal < 0 || signal >= (int)q->dep.size()) return false;
    lane = q->nmap > 0 ? q->map[lane % q->nmap] : lane % q->nq;
    if (!ring_room(q, lane)) return false;
    hsa_queue_t* hq = q->q[lane];
    hsa_barrier_and_packet_t* p = (hsa_barrier_and_packet_t*)hq->base_address + (q->written[lane] & (hq->size - 1));
    memset((char*)p + 2, 0, sizeof(*p) - 2);
    p->dep_signal[0] = q->dep[signal];
    // in order behind the lane's earlier packets; what the producing launch released (agent scope) is acquired by the launch behind
    const uint16_t header = (uint16_t)(HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (uint16_t)(1u << HSA_PACKET_HEADER_BARRIER) |
                            (uint16_t)(HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                            (uint16_t)(HSA_FENCE_SCOPE_NONE << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
    __atomic_store_n((uint16_t*)p, header, __ATOMIC_RELEASE);
    ++q->written[lane];
    hsa_queue_store_write_index_relaxed(hq, q->written[lane]);
    return true;
}

bool aql_dispatch(AqlQueue* q, int lane, const AqlKernel& k, uint32_t grid_wgs, uint32_t block_x, uint32_t dyn_lds, const void* kernarg, bool barrier, int done_signal) {
    lane = q->nmap > 0 ? q->map[lane % q->nmap] : lane % q->nq;
    hsa_queue_t* hq = q->q[lane];
    uint64_t& written = q->written[lane];
    if (!ring_room(q, lane)) return false;
    hsa_kernel_dispatch_packet_t* p = (hsa_kernel_dispatch_packet_t*)hq->base_address + (written & (hq->size - 1));
    p->setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
    p->workgroup_size_x = (uint16_t)block_x; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
    p->reserved0 = 0;
    p->grid_size_x = grid_wgs * block_x; p->grid_size_y = 1; p->grid_size_z = 1;
    p->private_segment_size = k.private_size;
    p->group_segment_size = k.group_static + dyn_lds;
    p->kernel_object = k.object;
    p->kernarg_address = const_cast<void*>(kernarg);
    p->reserved2 = 0;
    p->completion_signal.handle = (done_signal >= 0 && done_signal < (int)q->dep.size()) ? q->dep[done_signal].handle : 0;
    // agent-scope acquire at the start (L1 / scalar caches of the CUs), agent-scope release at the end; what a running consumer
    // reads of a running producer travels write-through and is read past the L1 (the kernels' business)
    uint16_t header = (uint16_t)(HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) |
                      (uint16_t)(HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                      (uint16_t)(HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
    if (barrier) header |= (uint16_t)(1u << HSA_PACKET_HEADER_BARRIER);
    // the header goes last: the packet processor may look at the slot as soon as the doorbell covers it
    __atomic_store_n((uint16_t*)p, header, __ATOMIC_RELEASE);
    ++written;
    hsa_queue_store_write_index_relaxed(hq, written);
    return true;
}

void aql_ring(AqlQueue* q) {
    for (int i = 0; i < q->nq; ++i) {
        if (q->rung[i] == q->written[i]) continue;
        hsa_signal_store_screlease(q->q[i]->doorbell_signal, (hsa_signal_value_t)(q->written[i] - 1));
        q->rung[i] = q->written[i];
    }
}

uint64_t aql_pending(AqlQueue* q) {
    uint64_t n = 0;
    for (int i = 0; i < q->nq; ++i) n += q->written[i] - hsa_queue_load_read_index_scacquire(q->q[i]);
    return n;
}

bool aql_wait_idle(AqlQueue* q, double timeout_s) {
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < q->nq; ++i) {
        hsa_queue_t* hq = q->q[i];
        while (q->written[i] - hsa_queue_load_read_index_scacquire(hq) >= hq->size) {
            aql_ring(q);
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
            std::this_thread::yield();
        }
        hsa_signal_store_relaxed(q->idle_sig[i], 1);
        hsa_barrier_and_packet_t* p = (hsa_barrier_and_packet_t*)hq->base_address + (q->written[i] & (hq->size - 1));
        memset((char*)p + 2, 0, sizeof(*p) - 2);
        p->completion_signal = q->idle_sig[i];
        const uint16_t header = (uint16_t)(HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (uint16_t)(1u << HSA_PACKET_HEADER_BARRIER) |
                                (uint16_t)(HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                                (uint16_t)(HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
        __atomic_store_n((uint16_t*)p, header, __ATOMIC_RELEASE);
        ++q->written[i];
        hsa_queue_store_write_index_relaxed(hq, q->written[i]);
    }
    aql_ring(q);
    uint64_t freq = 0;
    (void)hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &freq);
    const uint64_t ticks = freq ? (uint64_t)(timeout_s * (double)freq) : UINT64_MAX;
    bool ok = true;
    for (int i = 0; i < q->nq; ++i) {
        const hsa_signal_value_t v = hsa_signal_wait_scacquire(q->idle_sig[i], HSA_SIGNAL_CONDITION_LT, 1, ticks, HSA_WAIT_STATE_BLOCKED);
        if (v >= 1) ok = false;
    }
    std::lock_guard<std::mutex> lock(q->mu);
    if (!ok && q->error.empty()) q->error = "AQL queue did not drain in time";
    return ok && q->error.empty();
}

int aql_lanes(AqlQueue* q) { return q->nq; }
void aql_use_lanes(AqlQueue* q, int n, const int* hw) {
    q->nmap = 0;
    for (int i = 0; i < n && i < AqlQueue::MAXQ; ++i) if (hw[i] >= 0 && hw[i] < q->nq) q->map[q->nmap++] = hw[i];
}

std::string aql_state(AqlQueue* q) {
    char buf[512];
    std::string out;
    for (int i = 0; i < q->nq; ++i) {
        snprintf(buf, sizeof buf, "lane %d: written %llu read %llu; ", i, (unsigned long long)q->written[i], (unsigned long long)hsa_queue_load_read_index_scacquire(q->q[i]));
        out += buf;
    }
    out += "signals:";
    for (size_t i = 0; i < q->dep.size() && i < 6; ++i) { snprintf(buf, sizeof buf, " %lld", (long long)hsa_signal_load_relaxed(q->dep[i])); out += buf; }
    return out;
}

const char* aql_error(AqlQueue* q) {
    std::lock_guard<std::mutex> lock(q->mu);
    return q->error.c_str();
}

}  // namespace oph
