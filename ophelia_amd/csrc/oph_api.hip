// libophelia_hip.so -- model assembly, weight packing, decode loop and the C ABI
// (include/ophelia_hip.h).  Mirrors, for the synthesis path only:
//   networks.py  TextEnc 121-212, AudioEnc 214-284, Attention 286-325, AudioDec 360-435, SSRN 437-537
//   architectures.py 69-81, 139-142, 188-239 ; synthesize.py 150-260
// There is no CPU fallback anywhere in this file: every numeric result comes from the
// HIP kernels in oph_kernels.hip.
#include "oph_host.h"

#include <mutex>
#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>

thread_local std::string g_create_error;
thread_local hipStream_t g_cur = nullptr;    // stream the launch wrappers of THIS host thread target
thread_local int g_group_cls = -1;
double g_host_us[4] = {0, 0, 0, 0};     // OPH_TRACE: host time spent enqueuing {event ops, cone, critical launches, other}
const bool g_trace = getenv("OPH_TRACE") != nullptr;
thread_local std::string g_op_error;

// ====================================================================================== C ABI
extern "C" {

int oph_abi_version(void) { return OPH_ABI_VERSION; }

const char* oph_last_error(const oph_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

// The CU-masked streams of a device, process-wide: created on first use, never destroyed (see oph_create).
namespace {
struct MaskedSet { std::vector<uint32_t> key; hipStream_t s[3] = {nullptr, nullptr, nullptr}; int users = 0; };
std::mutex g_masked_mutex;
std::map<int, std::vector<MaskedSet>> g_masked;       // device -> sets (one per distinct partition; normally one)
std::map<int, std::recursive_mutex> g_device_mutex;   // device -> the lock under which the handles of a device share its masked streams
// Handles of one device SHARE the masked streams (a second set would put six masked queues on the device: time-slicing): whatever a
// call enqueues on them goes in under the device's lock (DevGuard at the API entry points), so the launches of two handles never
// interleave inside a decode -- each call's work is one contiguous run in stream order.  A handle that asks for a different CU
// partition than the process's first (CU_SPLIT sweeps) gets ordinary streams.
bool masked_streams_acquire(int device, int words, const uint32_t* m_dec, const uint32_t* m_conep, const uint32_t* m_ssrn,
                            hipStream_t* sdec, hipStream_t* scone, hipStream_t* sssrn) {
    std::lock_guard<std::mutex> lock(g_masked_mutex);
    std::vector<uint32_t> key;
    for (const uint32_t* m : {m_dec, m_conep, m_ssrn}) key.insert(key.end(), m, m + words);
    std::vector<MaskedSet>& sets = g_masked[device];
    // ONE set per device and process, for the process's life: masked queues are never destroyed (see oph_create), so a second set would
    // leave six of them on the device.  A handle that asks for another partition than the set's gets ordinary streams.
    if (sets.empty()) { sets.emplace_back(); sets[0].key = key; }
    MaskedSet& q = sets[0];
    if (q.key != key) return false;
    const uint32_t* masks[3] = {m_dec, m_conep, m_ssrn};
    for (int i = 0; i < 3; ++i)
        if (!q.s[i] && hipExtStreamCreateWithCUMask(&q.s[i], words, masks[i]) != hipSuccess) {
            (void)hipGetLastError();
            q.s[i] = nullptr;
            return false;                            // (the streams created so far are kept in the set and used by the next attempt)
        }
    q.users++;
    *sdec = q.s[0]; *scone = q.s[1]; *sssrn = q.s[2];
    return true;
}
void masked_streams_release(int device, hipStream_t sdec) {
    std::lock_guard<std::mutex> lock(g_masked_mutex);
    for (MaskedSet& q : g_masked[device])
        if (q.s[0] == sdec && q.users > 0) q.users--;
}
std::recursive_mutex& device_mutex(int device) {
    std::lock_guard<std::mutex> lock(g_masked_mutex);
    return g_device_mutex[device];
}
// Two PROCESSES of this library on one GPU (two ranks pointed at the same device: a test shape, or a mistake) would dispatch their
// whole-decode launches workgroup by workgroup onto the same CU partition, each get a part of it, and both run into the 2 s
// co-residency bound before the ladder saves them (round 5: 12 of 12 such runs).  Prevention instead of recovery: the calls that put
// work on the device take turns across processes too -- an advisory lock per physical GPU (flock on /dev/shm/ophelia_hip.<pci bus id>.lock),
// taken with the device's mutex and released with it.  One process per GPU never waits (two system calls per API call); a process
// that dies releases it; a holder that does not come back within 10 s is no longer waited for (the bounded in-kernel waits and the
// ladder are still there).  It is advisory: it orders THIS library's processes, not other tenants of the GPU.
struct GpuTurn { int fd = -2; int depth = 0; };          // fd -2: not opened yet, -1: unavailable
std::map<int, GpuTurn> g_gpu_turn;                     // (under the device's recursive mutex)
void gpu_turn_acquire(int device) {
    GpuTurn& t = g_gpu_turn[device];
    if (t.depth++ > 0) return;
    if (t.fd == -2) {
        char bus[64] = {0}, path[128];
        if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) { (void)hipGetLastError(); snprintf(bus, sizeof bus, "dev%d", device); }
        for (char* c = bus; *c; ++c) if (*c == ':' || *c == '/' || *c == '.') *c = '_';
        snprintf(path, sizeof path, "/dev/shm/ophelia_hip.%s.lock", bus);
        t.fd = open(path, O_CREAT | O_RDWR | O_CLOEXEC, 0666);
        if (t.fd >= 0) (void)fchmod(t.fd, 0666);
    }
    if (t.fd < 0) return;
    for (int i = 0; i < 50000; ++i) {                  // <= 10 s in 200 us naps
        if (flock(t.fd, LOCK_EX | LOCK_NB) == 0) return;
        if (errno != EWOULDBLOCK && errno != EINTR) return;
        struct timespec ts = {0, 200000};
        nanosleep(&ts, nullptr);
    }
    TRACE("another process has held this GPU's turn for 10 s: going on without it");
}
void gpu_turn_release(int device) {
    GpuTurn& t = g_gpu_turn[device];
    if (--t.depth > 0 || t.fd < 0) return;
    (void)flock(t.fd, LOCK_UN);
}
struct DevGuard {
    std::unique_lock<std::recursive_mutex> lk;
    int device = -1;
    explicit DevGuard(const oph_handle* h) {
        if (!h) return;
        lk = std::unique_lock<std::recursive_mutex>(device_mutex(h->device));
        device = h->device;
        gpu_turn_acquire(device);
    }
    ~DevGuard() { if (device >= 0) gpu_turn_release(device); }      // (before lk's destructor releases the mutex: members are destroyed after the body)
};
}  // namespace

int oph_create(const oph_dims* dims, int device, oph_handle** out) { return oph_create_opts(dims, device, nullptr, out); }

int oph_create_opts(const oph_dims* dims, int device, const char* options, oph_handle** out) {
    if (!dims || !out) { g_create_error = "null argument"; return OPH_ERR_INVALID; }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_create_error = "no HIP device available (libophelia_hip has no CPU fallback)";
        return OPH_ERR_DEVICE;
    }
    if (device < 0 || device >= ndev) { g_create_error = "device index out of range"; return OPH_ERR_INVALID; }
    const oph_dims& m = *dims;
    if (m.r != 4 && m.r != 8) { g_create_error = "reduction factor not handled by SSRN (networks.py:474-479)"; return OPH_ERR_UNSUPPORTED; }
    if (m.d % 4 || m.d > 256 || m.c % 4 || 2 * m.c > 1024 || m.n_mels > 256 || m.full_dim > 1280 || m.e % 4 ||
        m.attention_win_size < 1 || m.attention_win_size > 8 || m.max_N < 1 || m.max_T < 1 || m.vocab < 1) {
        g_create_error = "dimensions outside the supported hot path (d<=256, c<=512, n_mels<=256, full_dim<=1280, win<=8)";
        return OPH_ERR_UNSUPPORTED;
    }
    if ((m.flags & (OPH_FLAG_SPK_AUDIO_DECODER_INPUT | OPH_FLAG_SPK_TEXT_ENCODER_INPUT | OPH_FLAG_SPK_TEXT_ENCODER_TOWARDS_END | OPH_FLAG_LCC | OPH_FLAG_SPK_AUDIO_ENCODER_INPUT | OPH_FLAG_SPK_SSRN_INPUT)) &&
        (m.nspeakers < 1 || m.speaker_embedding_size < 1 || m.speaker_embedding_size % 4)) {
        g_create_error = "multispeaker flag set but nspeakers/speaker_embedding_size invalid";
        return OPH_ERR_INVALID;
    }
    if (hipSetDevice(device) != hipSuccess) { g_create_error = "hipSetDevice failed"; return OPH_ERR_DEVICE; }
    oph_handle* h = new oph_handle();
    h->dm = m;
    h->device = device;
    { std::string why; if (!h->opt.read(options, &why)) { g_create_error = why; delete h; return OPH_ERR_INVALID; } }
    static const char* names[PC_COUNT] = {"conv_gemm_f32<128,128>", "conv_gemm_f32<64,64>", "conv_gemm_bf16x3", "ln_rows", "dec_layer16", "row_chain", "attn_rows", "misc", "dec_run", "dec_loop", "cone_head", "hc_fused", "plane_gemm"};
    for (int i = 0; i < PC_COUNT; ++i) h->prof[i].name = names[i];
    // CU partition for the decode loop (MI355X: 256 CUs, mask bit i -> XCD i%8): the dependent layers of a step get a
    // private slice of 8 CUs in every XCD so the concurrently running history-cone GEMMs (the next 16 CUs per XCD) and the
    // SSRN partition (the last 8 per XCD) cannot delay them.
    {
        hipDeviceProp_t prop;
        uint32_t m_dec[16] = {0};
        uint32_t* m_cone = h->m_cone; uint32_t* m_conep = h->m_conep; uint32_t* m_ssrn = h->m_ssrn;
        int ncu = 0;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) ncu = prop.multiProcessorCount;
        const int words = (ncu + 31) / 32;
        int ndec = ncu / 4, nconep = ncu / 2;               // 64 | 128 | 64 of 256 CUs (sweep in DESIGN.md)
        if (h->opt.cu_dec > 0 && h->opt.cu_cone > 0 && h->opt.cu_dec + h->opt.cu_cone < ncu) { ndec = h->opt.cu_dec; nconep = h->opt.cu_cone; }
        if (ncu >= 64 && words <= 16 && !h->opt.no_cu_mask) {
            for (int i = 0; i < ncu; ++i) {
                const uint32_t bit = 1u << (i % 32);
                if (i < ndec) m_dec[i / 32] |= bit;
                else {
                    m_cone[i / 32] |= bit;
                    (i < ndec + nconep ? m_conep : m_ssrn)[i / 32] |= bit;
                }
            }
            // CU-masked queues are a scarce resource: with four alive the queues get time-sliced and even sequential batches
            // run 2x slower (measured), and re-creating masked streams after destroying some has hung hipStreamSynchronize.
            // So the three masked streams of a device (critical chain | cone | SSRN partitions) are created ONCE per process,
            // never destroyed, and SHARED by the handles of that device under the device's lock (masked_streams_acquire).
            // All three or none.
            h->mask_words = words;
            if (!masked_streams_acquire(device, words, m_dec, m_conep, m_ssrn, &h->sdec, &h->scone, &h->sssrn)) { h->sdec = h->scone = h->sssrn = nullptr; h->mask_words = 0; }
            else h->masked_borrowed = true;
            h->ndec_cus = h->mask_words ? ndec : ncu;
        } else h->ndec_cus = ncu;
        (void)hipGetLastError();
    }
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&h->scopy, hipStreamNonBlocking) != hipSuccess ||
        (!h->sdec && hipStreamCreateWithFlags(&h->sdec, hipStreamNonBlocking) != hipSuccess) ||
        (!h->scone && hipStreamCreateWithFlags(&h->scone, hipStreamNonBlocking) != hipSuccess) ||
        (!h->sssrn && hipStreamCreateWithFlags(&h->sssrn, hipStreamNonBlocking) != hipSuccess) ||
        hipEventCreateWithFlags(&h->ev_dec_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_ssrn_done[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_ssrn_done[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_preenc, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_copy, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_chunk, hipEventDisableTiming) != hipSuccess ||
        hipEventCreate(&h->ev_cs) != hipSuccess || hipEventCreate(&h->ev_ce) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_out, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_attn, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_cone, hipEventDisableTiming) != hipSuccess ||
        hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) {
        g_create_error = "stream/event creation failed";
        if (h->masked_borrowed) { masked_streams_release(h->device, h->sdec); h->sdec = h->scone = h->sssrn = nullptr; }
        delete h;
        return OPH_ERR_DEVICE;
    }
    g_cur = h->stream;
    {
        int can = 0;
        // Device words for the cross-stream dependencies.  The whole-decode launch (dec_loop) polls / raises them in-kernel.
        // For the per-step launch paths, stream write/wait-value operations on them are opt-in (OPH_STREAM_VALUE=1): under
        // rocprofv3 --pmc, which serialises dispatches across queues, a wait-value packet never sees the value the other queue
        // would write and the run deadlocks; events are understood by the profiler.
        h->d_sig = h->dalloc<uint32_t>(LOOP_SIG_WORDS);
        if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, device) == hipSuccess && can) {
            h->can_sigval = h->d_sig != nullptr && hipStreamWriteValue32(h->stream, h->d_sig, 0, 0) == hipSuccess &&
                            hipStreamSynchronize(h->stream) == hipSuccess;
            h->use_sigval = h->can_sigval && h->opt.stream_value;
        }
        (void)hipGetLastError();
        void* hp_ = nullptr;
        if (hipHostMalloc(&hp_, 64, hipHostMallocMapped) == hipSuccess) { h->host_prog = (volatile int*)hp_; h->host_prog[0] = -1; h->host_prog[1] = INT_MAX; }
        (void)hipGetLastError();
    }
    h->ssrn_prec = h->opt.ssrn_prec >= 0 ? h->opt.ssrn_prec : 2;
    build_networks(h);
    *out = h;
    return OPH_OK;
}

int oph_destroy(oph_handle* h) {
    DevGuard dev_guard(h);
    if (!h) return OPH_OK;
    hipSetDevice(h->device);
    TRACE("destroy: sync streams");
    for (hipStream_t st : {h->stream, h->sdec, h->scone, h->sssrn, h->scopy}) if (st) hipStreamSynchronize(st);
    for (auto& pc : h->prof)
        for (auto& e : pc.ev) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    for (hipEvent_t e : {h->ev0, h->ev1, h->ev_attn, h->ev_cone, h->ev_in, h->ev_out, h->ev_dec_done, h->ev_ssrn_done[0], h->ev_ssrn_done[1], h->ev_preenc, h->ev_copy, h->ev_chunk, h->ev_cs, h->ev_ce})
        if (e) hipEventDestroy(e);
    TRACE("destroy: free");
    h->free_pool(0); h->free_pool(1);
    if (h->host_prog) hipHostFree((void*)h->host_prog);
    TRACE("destroy: streams");
    if (h->masked_borrowed) {                     // the masked streams go back to the process-wide set (never destroyed)
        masked_streams_release(h->device, h->sdec);
        h->sdec = h->scone = h->sssrn = nullptr;
    }
    for (hipStream_t st : {h->scone, h->sssrn, h->sdec, h->scopy, h->stream}) if (st) { TRACE("  destroy stream %p", (void*)st); hipStreamDestroy(st); }
    TRACE("destroy: done");
    delete h;
    return OPH_OK;
}

// Pinned host memory for result buffers: device-to-host copies into it are true DMA (they overlap the running decode and
// reach the link's rate); any other host pointer is accepted everywhere too, at the cost of staged, blocking copies.
int oph_host_alloc(size_t bytes, void** out) {
    if (!out) return OPH_ERR_INVALID;
    *out = nullptr;
    if (hipHostMalloc(out, std::max<size_t>(bytes, 1), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return OPH_ERR_DEVICE; }
    return OPH_OK;
}
int oph_host_free(void* p) {
    if (!p) return OPH_OK;
    return hipHostFree(p) == hipSuccess ? OPH_OK : OPH_ERR_DEVICE;
}

static int check_ready(oph_handle* h, int B) {
    if (!h) return OPH_ERR_INVALID;
    if (!h->finalized) { h->fail("weights not finalized"); return OPH_ERR_STATE; }
    if (B < 1) { h->fail("batch must be >= 1"); return OPH_ERR_INVALID; }
    if (hipSetDevice(h->device) != hipSuccess) { h->fail("hipSetDevice failed"); return OPH_ERR_DEVICE; }
    return OPH_OK;
}
static bool model_is_multispeaker(const oph_handle* h) {
    return h->dm.flags & (OPH_FLAG_SPK_AUDIO_DECODER_INPUT | OPH_FLAG_SPK_TEXT_ENCODER_INPUT | OPH_FLAG_SPK_TEXT_ENCODER_TOWARDS_END | OPH_FLAG_LCC | OPH_FLAG_SPK_AUDIO_ENCODER_INPUT | OPH_FLAG_SPK_SSRN_INPUT);
}
// ends = get_text_lengths(L) (synthesize.py:242-247): a key position of the text, 0 ... max_N
static int check_ends(oph_handle* h, const int32_t* ends, int B) {
    for (int b = 0; b < B; ++b)
        if (ends[b] < 0 || ends[b] > h->dm.max_N) { h->fail("text end %d of utterance %d is outside the text (max_N = %d)", ends[b], b, h->dm.max_N); return OPH_ERR_INVALID; }
    return OPH_OK;
}
static int check_text(oph_handle* h, const int32_t* L, const int32_t* ends, const int32_t* spk, int B) {
    if (!L || !ends) { h->fail("null argument"); return OPH_ERR_INVALID; }
    if (int rc = check_ends(h, ends, B)) return rc;
    const bool ms = model_is_multispeaker(h);
    if (ms && !spk) { h->fail("multispeaker model needs speaker ids"); return OPH_ERR_INVALID; }
    const oph_dims& m = h->dm;
    for (long long i = 0; i < (long long)B * m.max_N; ++i)
        if (L[i] < 0 || L[i] >= m.vocab) { h->fail("text id %d out of range at %lld", L[i], i); return OPH_ERR_INVALID; }
    if (ms) for (int b = 0; b < B; ++b) if (spk[b] < 0 || spk[b] >= m.nspeakers) { h->fail("speaker id out of range"); return OPH_ERR_INVALID; }
    return OPH_OK;
}
// text of a batch into text buffer `slot`
static int upload_text(oph_handle* h, int slot, const int32_t* L, const int32_t* ends, const int32_t* spk, int B) {
    const oph_dims& m = h->dm;
    HIPCHK(h, hipMemcpyAsync(h->bL[slot], L, (size_t)B * m.max_N * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->bEnds[slot], ends, (size_t)B * 4, hipMemcpyHostToDevice, h->stream));
    if (model_is_multispeaker(h)) HIPCHK(h, hipMemcpyAsync(h->bSpk[slot], spk, (size_t)B * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return OPH_OK;
}

int oph_stage_text(oph_handle* h, const int32_t* L, const int32_t* ends, const int32_t* spk, int B) {
    DevGuard dev_guard(h);
    int rc = check_ready(h, B);
    if (rc) return rc;
    if ((rc = check_text(h, L, ends, spk, B))) return rc;
    if ((rc = ensure_decode_state(h, B))) return rc;
    if (h->preenc_valid) { HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_preenc, 0)); HIPCHK(h, hipStreamSynchronize(h->stream)); }    // a running pre-encode still reads the other slot
    h->preenc_valid = false; h->next_staged = false;      // a directly staged text supersedes a staged "next" one
    h->kv_resident = false; h->y_resident = false; h->txt_ran = false; h->kv_pre = false;
    if ((rc = upload_text(h, h->txt, L, ends, spk, B))) return rc;
    select_tile(h, 0);
    return OPH_OK;
}

// make the staged "next" text the current one (its K,V may already be there: pre-encoded under the previous decode)
static int advance_text(oph_handle* h) {
    if (!h->next_staged || !h->txt_ran) return OPH_OK;
    h->txt ^= 1; h->next_staged = false; h->txt_ran = false; h->kv_pre = false;
    h->kv_resident = false; h->y_resident = false;
    if (h->preenc_valid) {
        h->kv_cur ^= 1; h->preenc_valid = false; h->kv_pre = true;
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_preenc, 0));
    }
    select_tile(h, 0);
    return OPH_OK;
}

// The text of the batch AFTER the staged one, into the second text slot (same batch size: the tiles and buffers are
// shared).  While the staged batch decodes (oph_run_resident / oph_run_host) its TextEnc runs on the SSRN partition; the
// run after that switches to it and starts decoding at once.  If the staged batch has already run, this call makes the
// previously staged next text current first and then stages the new one behind it, so a caller simply alternates
//     oph_stage_text(t0); oph_stage_text_next(t1);  loop { run(); oph_stage_text_next(t_{i+2}); }
int oph_stage_text_next(oph_handle* h, const int32_t* L, const int32_t* ends, const int32_t* spk, int B) {
    DevGuard dev_guard(h);
    int rc = check_ready(h, B);
    if (rc) return rc;
    if (!h->bKV[0]) { h->fail("stage a first batch with oph_stage_text"); return OPH_ERR_STATE; }
    if (B != h->nB) { h->fail("the next batch must have the staged batch's size (%d)", h->nB); return OPH_ERR_INVALID; }
    if ((rc = check_text(h, L, ends, spk, B))) return rc;
    if ((rc = advance_text(h))) return rc;
    if (h->next_staged) { h->fail("a next batch is already staged and the current one has not run yet"); return OPH_ERR_STATE; }
    if ((rc = upload_text(h, h->txt ^ 1, L, ends, spk, B))) return rc;
    h->next_staged = true; h->next_B = B;
    return OPH_OK;
}

int oph_decode_steps(oph_handle* h, int t_begin, int t_end, int stop_mode, int32_t* steps_run) {
    DevGuard dev_guard(h);
    if (!h || !h->bKV[0]) { if (h) h->fail("no staged batch"); return OPH_ERR_STATE; }
    if (stop_mode != OPH_STOP_REFERENCE && stop_mode != OPH_STOP_NEVER) { h->fail("unknown stop mode %d", stop_mode); return OPH_ERR_INVALID; }
    if (t_begin < 0 || t_end < t_begin || t_end > h->dm.max_T) { h->fail("steps [%d, %d) are outside [0, max_T = %d]", t_begin, t_end, h->dm.max_T); return OPH_ERR_INVALID; }
    HIPCHK(h, hipSetDevice(h->device));
    if (t_begin == 0) { begin_batch(h); return decode_batch(h, t_end, stop_mode, steps_run); }
    // resume (multi-GPU global stop): every tile continues from the step the batch stopped at; clear the local stops
    const int ntiles = (h->nB + TILE - 1) / TILE;
    int last = t_begin;
    int back = 0, ahead = 0;
    ssrn_margins(h, &back, &ahead);
    for (int j = 0; j < ntiles; ++j) {
        select_tile(h, j);
        if (h->tiles[j].steps != t_begin) { h->fail("resume at step %d, but tile %d stands at step %d", t_begin, j, h->tiles[j].steps); return OPH_ERR_STATE; }
        const int ctl1 = INT_MAX;
        HIPCHK(h, hipMemcpyAsync(h->d_ctl + 1, &ctl1, 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        h->tiles[j].ssrn_done = std::min(h->tiles[j].ssrn_done, std::max(0, t_begin - ahead));      // frames >= t_begin change: SSRN rows that saw them are stale
        h->tiles[j].z_copied = std::min(h->tiles[j].z_copied, h->tiles[j].ssrn_done);
        int32_t st = 0;
        const int rc = decode_range(h, t_begin, t_end, stop_mode, &st);
        if (rc) return rc;
        last = std::max(last, (int)st);
    }
    select_tile(h, 0);
    if (steps_run) *steps_run = last;
    return OPH_OK;
}

// Switch between sequential batches (everything joined when a call returns) and pipelined batches (the SSRN tail of a
// batch stays on the SSRN partition and overlaps the next batch).
static int set_pipelined(oph_handle* h, bool pipe) {
    if (pipe == h->pipelined) return OPH_OK;
    for (hipStream_t st : {h->stream, h->sdec, h->scone, h->sssrn, h->scopy}) if (st) HIPCHK(h, hipStreamSynchronize(st));
    if (!pipe) h->buf = 0;
    h->pipelined = pipe;
    h->ssrn_inflight[0] = h->ssrn_inflight[1] = false;
    select_tile(h, h->tile);
    return OPH_OK;
}

int oph_run_ssrn_resident(oph_handle* h) {
    DevGuard dev_guard(h);
    if (!h || !h->bKV[0]) { if (h) h->fail("no staged batch"); return OPH_ERR_STATE; }
    HIPCHK(h, hipSetDevice(h->device));
    g_cur = h->stream;
    return finish_ssrn(h);
}

int oph_run_resident(oph_handle* h, int stop_mode, int run_ssrn, int32_t* steps_run) {
    DevGuard dev_guard(h);
    if (!h || !h->bKV[0]) { if (h) h->fail("no staged batch"); return OPH_ERR_STATE; }
    if (stop_mode != OPH_STOP_REFERENCE && stop_mode != OPH_STOP_NEVER) { h->fail("unknown stop mode %d", stop_mode); return OPH_ERR_INVALID; }
    if (run_ssrn < 0 || run_ssrn > 2) { h->fail("run_ssrn must be 0, 1 or 2"); return OPH_ERR_INVALID; }
    HIPCHK(h, hipSetDevice(h->device));
    // run_ssrn: 0 = Text2Mel only; 1 = SSRN too, joined when the call returns; 2 = pipelined batches -- the SSRN tail of
    // THIS batch (what its streamed chunks have not covered when the decode ends) stays queued on the SSRN partition and
    // overlaps the next call's decode (Y/Z ping-pong); oph_synchronize / oph_fetch_* / oph_timer_stop join it.
    const bool pipe = run_ssrn == 2;
    int rc = set_pipelined(h, pipe);
    if (rc) return rc;
    g_cur = h->stream;
    if ((rc = advance_text(h))) return rc;       // the current text has run: a text staged with oph_stage_text_next takes its place
    begin_batch(h);
    select_tile(h, 0);
    if (h->kv_pre) h->n_preenc_used++;
    else if ((rc = run_encode(h))) return rc;
    h->kv_pre = false; h->txt_ran = true;
    h->kv_resident = true;
    h->want_preenc = !h->opt.no_preencode;
    const bool spec_saved = h->spec_ssrn;
    h->spec_ssrn = run_ssrn != 0;
    rc = decode_batch(h, h->dm.max_T, stop_mode, steps_run);
    h->want_preenc = false; h->spec_ssrn = spec_saved;
    if (rc) return rc;
    h->y_resident = true;
    if (run_ssrn) rc = finish_ssrn(h);
    return rc;
}

int oph_set_ssrn_precision(oph_handle* h, int mode) {
    if (!h || mode < 0 || mode > 4) return OPH_ERR_INVALID;       // 3, 4: measurement only -- split-fp16 with 2 products / 1 product
    if (h->guard_ssrn && mode != 0) { h->fail("an SSRN weight is outside fp16's range: only the fp32-operand MFMA (mode 0) is offered"); return OPH_ERR_UNSUPPORTED; }
    h->ssrn_prec = mode; h->chunk_ms = 0.f;
    return OPH_OK;
}
// which: 0 SSRN (= oph_set_ssrn_precision), 1 the cone's two many-row contractions, 2 TextEnc.  mode: 0 fp32 MFMA, 2 split-fp16 x3
// (fp32-class), 1 split-bf16 x3 (SSRN; the cone only if the handle was created under OPH_CONE_PREC=1)
int oph_set_precision(oph_handle* h, int which, int mode) {
    if (!h || mode < 0 || mode > 2) return OPH_ERR_INVALID;
    if (mode != 0 && ((which == 0 && h->guard_ssrn) || (which == 1 && h->guard_cone) || (which == 2 && h->guard_text))) {
        h->fail("a weight of that net is outside fp16's range: only the fp32-operand MFMA (mode 0) is offered"); return OPH_ERR_UNSUPPORTED;
    }
    if (which == 0) { h->ssrn_prec = mode; h->chunk_ms = 0.f; return OPH_OK; }
    if (which == 1) {
        if (mode == 1 && !(h->n_hc_dec > 1 && h->audiodec[h->dec_pre].Wh)) { h->fail("the cone's bf16 planes were not built (create the handle with the option CONE_PREC=1)"); return OPH_ERR_STATE; }
        h->cone_prec = mode; return OPH_OK;
    }
    if (which == 2 && mode != 1) { h->textenc_prec = mode; return OPH_OK; }
    h->fail("bad precision selector");
    return OPH_ERR_INVALID;
}
// what the pipeline actually did since the handle was created (tests and bench.py assert on these):
// [0] TextEnc evaluations  [1] runs that found their K,V pre-encoded  [2] SSRN chunks launched while a decode was running
// [3] whole-decode launches  [4] fallbacks from the whole-decode launch to two launches per step  [5] tiles resumed to the batch's stop step
int oph_get_counters(oph_handle* h, int64_t* out, int n) {
    if (!h || !out) return OPH_ERR_INVALID;
    const long long v[11] = {h->n_textenc, h->n_preenc_used, h->n_chunks_streamed, h->n_loop_decodes, h->n_loop_fallbacks, h->n_tile_resumes, 0LL /* (was: persistent cone launches; removed in round 6) */,
                             (long long)((h->guard_ssrn ? 1 : 0) | (h->guard_cone ? 2 : 0) | (h->guard_text ? 4 : 0)),
                             h->mask_words > 0 ? 1 : 0, h->n_recoveries, h->degraded_left};
    for (int i = 0; i < n && i < 11; ++i) out[i] = v[i];
    return OPH_OK;
}
// Where the speculative SSRN of the NEXT oph_text2mel copies its rows while the decoder is still running: a host buffer of
// (B, r*max_T, full_dim) floats (pinned, oph_host_alloc, for the copies to be asynchronous).  oph_ssrn(Y = NULL, ..., Z = that
// pointer) then only has the tail left to compute and copy.  NULL clears it.
int oph_set_mag_destination(oph_handle* h, float* Z) {
    if (!h) return OPH_ERR_INVALID;
    h->z_spec = Z; h->z_spec_gen = 0;
    return OPH_OK;
}
int oph_set_streaming(oph_handle* h, int on) {
    if (!h || on < 0) return OPH_ERR_INVALID;
    h->spec_ssrn = on != 0;
    if (on >= 2) h->opt.ssrn_chunk = on;          // mel frames per streamed chunk (1 keeps the current size)
    return OPH_OK;
}

int oph_synchronize(oph_handle* h) {
    DevGuard dev_guard(h);
    if (!h) return OPH_ERR_INVALID;
    if (h->sssrn) HIPCHK(h, hipStreamSynchronize(h->sssrn));
    if (h->scopy) HIPCHK(h, hipStreamSynchronize(h->scopy));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    return OPH_OK;
}

int oph_fetch_kv(oph_handle* h, float* K, float* V) {
    DevGuard dev_guard(h);
    if (!h || !h->bKV[0]) { if (h) h->fail("no staged batch"); return OPH_ERR_STATE; }
    const oph_dims& m = h->dm;
    const size_t rows = (size_t)h->nB * m.max_N, w = (size_t)m.d * 4;
    const float* kv = h->bKV[h->kv_cur];
    if (K) HIPCHK(h, hipMemcpy2DAsync(K, w, kv, 2 * w, w, rows, hipMemcpyDeviceToHost, h->stream));
    if (V) HIPCHK(h, hipMemcpy2DAsync(V, w, kv + m.d, 2 * w, w, rows, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return OPH_OK;
}

int oph_fetch_mel(oph_handle* h, float* Y, int32_t* t_ends, float* alignments) {
    DevGuard dev_guard(h);
    if (!h || !h->bKV[0]) { if (h) h->fail("no staged batch"); return OPH_ERR_STATE; }
    const oph_dims& m = h->dm;
    if (Y) HIPCHK(h, hipMemcpy2DAsync(Y, (size_t)m.n_mels * 4, h->bYout[h->buf], (size_t)h->ldy * 4, (size_t)m.n_mels * 4,
                                      (size_t)h->nB * m.max_T, hipMemcpyDeviceToHost, h->stream));
    if (t_ends) HIPCHK(h, hipMemcpyAsync(t_ends, h->bTends, (size_t)h->nB * 4, hipMemcpyDeviceToHost, h->stream));
    if (alignments) HIPCHK(h, hipMemcpyAsync(alignments, h->bAlign, (size_t)h->nB * m.max_N * m.max_T * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return OPH_OK;
}

int oph_fetch_mag(oph_handle* h, float* Z) {
    DevGuard dev_guard(h);
    if (!h || !h->bKV[0] || !Z) { if (h) h->fail("no staged batch / null"); return OPH_ERR_STATE; }
    if (h->sssrn) HIPCHK(h, hipStreamSynchronize(h->sssrn));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    { const int rc = check_convt_ln(h); if (rc) return rc; }
    const oph_dims& m = h->dm;
    HIPCHK(h, hipMemcpyAsync(Z, h->bZ[h->buf], (size_t)h->nB * m.max_T * m.r * m.full_dim * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return OPH_OK;
}

int oph_device_mag(oph_handle* h, const float** d_mag, int64_t* utt_stride, int32_t* B) {
    if (!h || !h->bKV[0] || !d_mag) { if (h) h->fail("no staged batch / null"); return OPH_ERR_STATE; }
    if (h->sssrn) HIPCHK(h, hipStreamSynchronize(h->sssrn));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    { const int rc = check_convt_ln(h); if (rc) return rc; }
    const oph_dims& m = h->dm;
    *d_mag = h->bZ[h->buf];
    if (utt_stride) *utt_stride = (int64_t)m.max_T * m.r * m.full_dim;
    if (B) *B = h->nB;
    return OPH_OK;
}

// One whole batch host -> host: the staged text (oph_stage_text, or the one staged with oph_stage_text_next during the
// previous call) through TextEnc, decode and SSRN, with every result copied into the caller's buffers (any of them may be
// NULL) as it becomes final -- SSRN rows chunk by chunk and Y / alignments at the end of the decode, on the copy stream, under
// the work that is still running.  Pinned buffers (oph_host_alloc) make those copies asynchronous DMA.  What the
// reference's two clocks bracket (synthesize.py:553-576).
int oph_run_host(oph_handle* h, int stop_mode, float* K, float* V, float* Y, int32_t* t_ends, float* alignments, float* Z, int32_t* steps_run) {
    DevGuard dev_guard(h);
    if (!h || !h->bKV[0]) { if (h) h->fail("no staged batch"); return OPH_ERR_STATE; }
    if (stop_mode != OPH_STOP_REFERENCE && stop_mode != OPH_STOP_NEVER) { h->fail("unknown stop mode %d", stop_mode); return OPH_ERR_INVALID; }
    HIPCHK(h, hipSetDevice(h->device));
    int rc = set_pipelined(h, false);
    if (rc) return rc;
    const oph_dims& m = h->dm;
    using clk = std::chrono::steady_clock;
    const clk::time_point tp0 = clk::now();
    clk::time_point tp1 = tp0, tp2 = tp0, tp3 = tp0;
    g_cur = h->stream;
    if ((rc = advance_text(h))) return rc;
    begin_batch(h);
    select_tile(h, 0);
    if (h->kv_pre) h->n_preenc_used++;
    else if ((rc = run_encode(h))) return rc;
    h->kv_pre = false; h->txt_ran = true;
    h->kv_resident = true;
    if (K || V) {       // K,V leave on the copy stream while the decode starts
        HIPCHK(h, hipEventRecord(h->ev_copy, h->stream));
        HIPCHK(h, hipStreamWaitEvent(h->scopy, h->ev_copy, 0));
        const size_t rows = (size_t)h->nB * m.max_N, w = (size_t)m.d * 4;
        const float* kv = h->bKV[h->kv_cur];
        if (K) HIPCHK(h, hipMemcpy2DAsync(K, w, kv, 2 * w, w, rows, hipMemcpyDeviceToHost, h->scopy));
        if (V) HIPCHK(h, hipMemcpy2DAsync(V, w, kv + m.d, 2 * w, w, rows, hipMemcpyDeviceToHost, h->scopy));
    }
    h->want_preenc = !h->opt.no_preencode;
    const bool spec_saved = h->spec_ssrn;
    h->spec_ssrn = Z != nullptr;
    h->z_host = Z;
    tp1 = clk::now();
    rc = decode_batch(h, m.max_T, stop_mode, steps_run);
    tp2 = clk::now();
    h->want_preenc = false; h->spec_ssrn = spec_saved;
    if (rc) { h->z_host = nullptr; return rc; }
    h->y_resident = true;
    // Y, t_ends, alignments: final now (the API stream has joined the decode streams)
    HIPCHK(h, hipEventRecord(h->ev_copy, h->stream));
    HIPCHK(h, hipStreamWaitEvent(h->scopy, h->ev_copy, 0));
    if (Y) HIPCHK(h, hipMemcpy2DAsync(Y, (size_t)m.n_mels * 4, h->bYout[h->buf], (size_t)h->ldy * 4, (size_t)m.n_mels * 4, (size_t)h->nB * m.max_T, hipMemcpyDeviceToHost, h->scopy));
    if (t_ends) HIPCHK(h, hipMemcpyAsync(t_ends, h->bTends, (size_t)h->nB * 4, hipMemcpyDeviceToHost, h->scopy));
    if (alignments) HIPCHK(h, hipMemcpyAsync(alignments, h->bAlign, (size_t)h->nB * m.max_N * m.max_T * 4, hipMemcpyDeviceToHost, h->scopy));
    if (Z) rc = finish_ssrn(h);
    h->z_host = nullptr;
    if (rc) return rc;
    tp3 = clk::now();
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipStreamSynchronize(h->sssrn));
    HIPCHK(h, hipStreamSynchronize(h->scopy));
    if (g_trace) {
        auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        TRACE("run_host: text / K,V %.3f ms, decode_batch %.3f ms, finish_ssrn enqueue %.3f ms, final syncs %.3f ms", ms(tp0, tp1), ms(tp1, tp2), ms(tp2, tp3), ms(tp3, clk::now()));
    }
    return check_convt_ln(h, Z);
}

// ---- host-buffer session calls ---------------------------------------------------------------
int oph_encode_text(oph_handle* h, const int32_t* L, const int32_t* spk, int B, float* K, float* V) {
    DevGuard dev_guard(h);
    int rc = check_ready(h, B);
    if (rc) return rc;
    if (!L || !K || !V) { h->fail("null argument"); return OPH_ERR_INVALID; }
    // ends are not needed by TextEnc; stage zeros (get_text_lengths stays with the caller, synthesize.py:556)
    std::vector<int32_t> ends(B, 0), spk0(B, 0);
    if ((rc = oph_stage_text(h, L, ends.data(), spk ? spk : spk0.data(), B))) return rc;
    g_cur = h->stream;
    if ((rc = run_encode(h))) return rc;
    h->kv_resident = true;          // oph_text2mel(K = NULL, V = NULL) decodes from these
    return oph_fetch_kv(h, K, V);
}

// shared front of the two decode entry points: K,V (or the resident ones), ends / speakers of the batch
static int stage_decode_inputs(oph_handle* h, const float* K, const float* V, bool need_k, const int32_t* ends, const int32_t* spk, int B) {
    int rc;
    const bool ms = h->dm.flags & (OPH_FLAG_SPK_AUDIO_DECODER_INPUT | OPH_FLAG_LCC | OPH_FLAG_SPK_AUDIO_ENCODER_INPUT);
    if (ms && !spk) { h->fail("multispeaker model needs speaker ids"); return OPH_ERR_INVALID; }
    const oph_dims& m = h->dm;
    if (!K && !V) {
        if (!h->kv_resident || B != h->nB) { h->fail("K = V = NULL asks for the K,V oph_encode_text left in HBM, but there are none for a batch of %d", B); return OPH_ERR_STATE; }
    } else {
        if ((need_k && !K) || !V) { h->fail("null argument"); return OPH_ERR_INVALID; }
        if ((rc = ensure_decode_state(h, B))) return rc;
        const size_t rows = (size_t)B * m.max_N, w = (size_t)m.d * 4;
        float* kv = h->bKV[h->kv_cur];
        if (K) HIPCHK(h, hipMemcpy2DAsync(kv, 2 * w, K, w, w, rows, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpy2DAsync(kv + m.d, 2 * w, V, w, w, rows, hipMemcpyHostToDevice, h->stream));
        h->kv_resident = false;
    }
    if (ends) {
        if ((rc = check_ends(h, ends, B))) return rc;
        HIPCHK(h, hipMemcpyAsync(h->bEnds[h->txt], ends, (size_t)B * 4, hipMemcpyHostToDevice, h->stream));
    }
    if (ms) {
        for (int b = 0; b < B; ++b) if (spk[b] < 0 || spk[b] >= m.nspeakers) { h->fail("speaker id out of range"); return OPH_ERR_INVALID; }
        HIPCHK(h, hipMemcpyAsync(h->bSpk[h->txt], spk, (size_t)B * 4, hipMemcpyHostToDevice, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return OPH_OK;
}

int oph_text2mel(oph_handle* h, const float* K, const float* V, const int32_t* ends, const int32_t* spk, int B,
                 int stop_mode, float* Y, int32_t* t_ends, float* alignments, int32_t* steps_run) {
    DevGuard dev_guard(h);
    int rc = check_ready(h, B);
    if (rc) return rc;
    if (!ends) { h->fail("null argument"); return OPH_ERR_INVALID; }
    if (stop_mode != OPH_STOP_REFERENCE && stop_mode != OPH_STOP_NEVER) { h->fail("unknown stop mode %d", stop_mode); return OPH_ERR_INVALID; }
    if ((rc = set_pipelined(h, false))) return rc;
    if ((rc = stage_decode_inputs(h, K, V, true, ends, spk, B))) return rc;
    begin_batch(h);
    h->z_host = h->spec_ssrn ? h->z_spec : nullptr;       // the speculative SSRN's rows leave for the host as they are produced
    h->z_spec_gen = h->z_host ? h->batch_gen : 0;         // this batch's rows are the ones in z_spec
    rc = decode_batch(h, h->dm.max_T, stop_mode, steps_run);
    h->z_host = nullptr;
    if (rc) return rc;
    h->y_resident = true;           // oph_ssrn(Y = NULL) continues from here
    return oph_fetch_mel(h, Y, t_ends, alignments);
}

int oph_text2mel_durations(oph_handle* h, const float* K, const float* V, const float* durations, const int32_t* spk,
                           int B, int n_steps, float* Y, int32_t* t_ends, float* alignments, int32_t* steps_run) {
    DevGuard dev_guard(h);
    int rc = check_ready(h, B);
    if (rc) return rc;
    if (!durations) { h->fail("null argument"); return OPH_ERR_INVALID; }
    if ((rc = set_pipelined(h, false))) return rc;
    if ((rc = stage_decode_inputs(h, K, V, false, nullptr, spk, B))) return rc;
    const oph_dims& m = h->dm;
    const int ntiles = (B + TILE - 1) / TILE;
    // selection matrix -> key index per (t, b); only hard 0/1 rows (what data_load.py:243-251 produces) are supported
    std::vector<std::vector<int>> ptab(ntiles, std::vector<int>((size_t)m.max_T * TILE, -1));
    std::vector<int32_t> tends(B, 0);
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < m.max_T; ++t) {
            const float* row = durations + ((size_t)b * m.max_T + t) * m.max_N;
            int key = -1;
            for (int n = 0; n < m.max_N; ++n) {
                if (row[n] == 0.0f) continue;
                if (row[n] != 1.0f || key >= 0) { h->fail("durations row (b=%d, t=%d) is not a 0/1 selection of at most one key", b, t); return OPH_ERR_UNSUPPORTED; }
                key = n;
            }
            if (key >= 0) { ptab[b / TILE][(size_t)t * TILE + b % TILE] = key; tends[b]++; }
        }
    int steps = n_steps;
    if (steps <= 0) {
        int mx = 0;
        for (int b = 0; b < B; ++b) mx = std::max(mx, (int)tends[b]);
        steps = std::min((int)m.max_T, mx + 1);          // synthesize.py:211-216: the step at which j >= max(t_ends) still runs
    }
    steps = std::min(steps, (int)m.max_T);
    for (int j = 0; j < ntiles; ++j)
        HIPCHK(h, hipMemcpyAsync(h->tiles[j].d_ptab, ptab[j].data(), ptab[j].size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    begin_batch(h);
    h->fixed_att = true;
    rc = decode_batch(h, steps, OPH_STOP_NEVER, nullptr);
    h->fixed_att = false;
    if (rc) return rc;
    h->y_resident = true;
    if ((rc = oph_fetch_mel(h, Y, nullptr, alignments))) return rc;
    if (t_ends) std::copy(tends.begin(), tends.end(), t_ends);
    if (steps_run) *steps_run = steps;
    return OPH_OK;
}

// One evaluation of the synthesis graph at fixed feeds -- what ONE sess.run of the reference's loop computes
// (synthesize.py:181-183 with the feed dict of :172): S = shift(mels) -> AudioEnc over all max_T positions ->
// Attention of every position under the ONE mask prev_max (networks.py:300-319) -> AudioDec -> logits / sigmoid.
// O(max_T) work per call: the debug / fetch surface of architectures.py:188-239 (g.Q, g.R, g.Y_logits, ...), not the
// decode loop (oph_text2mel).  Batched kernels: conv_gemm_f32 + ln_rows per layer, attn_rows.
int oph_text2mel_graph(oph_handle* h, const float* K, const float* V, const float* mels, const int32_t* prev_max,
                       const int32_t* ends, const int32_t* spk, int B,
                       float* Q, float* R, float* Y_logits, float* Y, float* alignments, int32_t* max_attentions) {
    DevGuard dev_guard(h);
    int rc = check_ready(h, B);
    if (rc) return rc;
    if (!K || !V || !mels || !prev_max) { h->fail("null argument"); return OPH_ERR_INVALID; }
    const oph_dims& m = h->dm;
    if ((m.flags & OPH_FLAG_NO_MONOTONIC) && !ends) { h->fail("turn_off_monotonic_for_synthesis needs the text lengths"); return OPH_ERR_INVALID; }
    for (int b = 0; b < B; ++b) if (prev_max[b] < 0 || prev_max[b] >= m.max_N) { h->fail("prev_max_attentions out of range"); return OPH_ERR_INVALID; }
    if ((rc = set_pipelined(h, false))) return rc;
    if ((rc = stage_decode_inputs(h, K, V, true, ends, spk, B))) return rc;
    h->y_resident = false;
    g_cur = h->stream;
    const int T = m.max_T, d = m.d, ldy = h->ldy;
    // the fed prev_max_attentions of all utterances: one int per utterance, batch order (bTends doubles as the feed buffer)
    int* d_pm = h->bTends;
    HIPCHK(h, hipMemcpyAsync(d_pm, prev_max, (size_t)B * 4, hipMemcpyHostToDevice, h->stream));
    // S = concat(zeros, mels[:, :-1])  (architectures.py:191), rows padded to the first layer's K
    std::vector<float> S((size_t)B * T * ldy, 0.f);
    for (int b = 0; b < B; ++b)
        for (int t = 1; t < T; ++t)
            std::copy(mels + ((size_t)b * T + t - 1) * m.n_mels, mels + ((size_t)b * T + t) * m.n_mels, S.begin() + ((size_t)b * T + t) * ldy);
    HIPCHK(h, hipMemcpyAsync(h->actA, S.data(), S.size() * 4, hipMemcpyHostToDevice, h->stream));
    int ldq = 0;
    float* dQ = run_batched(h, h->audioenc, h->actA, ldy, B, T, 0, 0, nullptr, 0, 0, &ldq, nullptr);
    // attention over all positions, one mask per utterance; R' rows go to the workspace the encoder did not end in
    float* dR = dQ == h->actA ? h->actB : h->actA;
    const float* kv = h->bKV[h->kv_cur];
    HIPCHK(h, hipMemsetAsync(h->bAlign, 0, (size_t)h->nBpad * m.max_N * T * 4, h->stream));
    AttnRowsArgs a{};
    a.mode = 1; a.Q = dQ; a.ldq = ldq; a.K = kv; a.V = kv + d; a.ldkv = 2 * d; a.N = m.max_N; a.d = d; a.win = m.attention_win_size;
    a.p = d_pm; a.B = B; a.Bpad = h->nBpad; a.nrows = B * T; a.T = T; a.R = dR; a.ldr = 2 * d; a.align = h->bAlign; a.amax = h->d_amax;
    if (m.flags & OPH_FLAG_NO_MONOTONIC) a.ends = h->bEnds[h->txt];
    launch_attn_rows(a, h->stream);
    if (Q) HIPCHK(h, hipMemcpy2DAsync(Q, (size_t)d * 4, dQ, (size_t)ldq * 4, (size_t)d * 4, (size_t)B * T, hipMemcpyDeviceToHost, h->stream));
    if (R) HIPCHK(h, hipMemcpyAsync(R, dR, (size_t)B * T * 2 * d * 4, hipMemcpyDeviceToHost, h->stream));
    if (alignments) HIPCHK(h, hipMemcpyAsync(alignments, h->bAlign, (size_t)B * m.max_N * T * 4, hipMemcpyDeviceToHost, h->stream));
    std::vector<long long> amax((size_t)B * T);
    HIPCHK(h, hipMemcpyAsync(amax.data(), h->d_amax, amax.size() * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));     // Q may live in the buffer the decoder overwrites next
    // AudioDec; its last layer's LayerNorm rows are stored twice: squashed (g.Y) and as they are (g.Y_logits)
    int ldl = 0;
    BatchedIO io{};
    float* dlogits = h->raw;         // free once the last layer's epilogues have run (they read it row by row: use the far end)
    const int ldn = round_up(m.n_mels, 32);
    dlogits = h->raw + (h->raw_elems - (size_t)B * T * ldn);
    io.final_logits = dlogits;
    std::vector<Layer> dec = h->audiodec;
    dec.back().act = (m.flags & OPH_FLAG_NO_SQUASH_T2M) ? ACT_NONE : ACT_SIGMOID;                   // squash_output_t2m (networks.py:430-433)
    float* dY = run_batched(h, dec, dR, 2 * d, B, T, 0, 0, nullptr, 0, 0, &ldl, nullptr, io);
    if (Y) HIPCHK(h, hipMemcpy2DAsync(Y, (size_t)m.n_mels * 4, dY, (size_t)ldl * 4, (size_t)m.n_mels * 4, (size_t)B * T, hipMemcpyDeviceToHost, h->stream));
    if (Y_logits) HIPCHK(h, hipMemcpy2DAsync(Y_logits, (size_t)m.n_mels * 4, dlogits, (size_t)ldl * 4, (size_t)m.n_mels * 4, (size_t)B * T, hipMemcpyDeviceToHost, h->stream));
    hipError_t e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess || (e = hipGetLastError()) != hipSuccess) { h->fail("graph evaluation failed: %s", hipGetErrorString(e)); return OPH_ERR_DEVICE; }
    if (max_attentions) for (size_t i = 0; i < amax.size(); ++i) max_attentions[i] = (int32_t)amax[i];
    // bTends was borrowed for the fed prev_max: put the "not ended" state back
    launch_fill_int(h->bTends, m.max_T, h->nBpad, h->stream);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return OPH_OK;
}

// oph_ssrn / oph_ssrn_logits.  Y == NULL: the mel frames the last decode call left in HBM (B and T must be that batch's);
// whatever its streamed SSRN has not covered yet is computed now.
static int ssrn_common(oph_handle* h, const float* Y, int B, int T, float* Z, float* Z_logits, const int32_t* spk = nullptr) {
    DevGuard dev_guard(h);
    int rc = check_ready(h, B);
    if (rc) return rc;
    if (!Z || T < 1) { h->fail("bad argument"); return OPH_ERR_INVALID; }
    const oph_dims& m = h->dm;
    if (spk && !(m.flags & OPH_FLAG_SPK_SSRN_INPUT)) { h->fail("speaker codes given, but 'ssrn_input' is not in this configuration's multispeaker positions"); return OPH_ERR_INVALID; }
    if (Y && !spk && (m.flags & OPH_FLAG_SPK_SSRN_INPUT)) { h->fail("'ssrn_input': the SSRN graph needs the speaker codes (oph_ssrn_speakers) -- g.speakers is not fed"); return OPH_ERR_INVALID; }
    if (spk) for (int i = 0; i < B; ++i) if (spk[i] < 0 || spk[i] >= m.nspeakers) { h->fail("speaker id %d out of range", spk[i]); return OPH_ERR_INVALID; }
    if (T > m.max_T) { h->fail("T=%d exceeds max_T=%d", T, m.max_T); return OPH_ERR_INVALID; }
    if (!Y && !Z_logits) {
        if (!h->y_resident || B != h->nB || T != m.max_T) { h->fail("Y = NULL asks for the mel frames the last decode left in HBM, but there are none for B=%d, T=%d", B, T); return OPH_ERR_STATE; }
        g_cur = h->stream;
        if (Z == h->z_spec && Z != nullptr && h->z_spec_gen == h->batch_gen) {
            // the chunks streamed during THIS batch's decode are already in Z (oph_set_mag_destination; the batch generation tells a
            // later batch on the same handle apart): compute and copy what is left -- every row past the copied frontier
            h->z_host = Z;
            rc = finish_ssrn(h);
            h->z_host = nullptr;
            if (rc) return rc;
            HIPCHK(h, hipStreamSynchronize(h->stream));
            HIPCHK(h, hipStreamSynchronize(h->sssrn));
            HIPCHK(h, hipStreamSynchronize(h->scopy));
            return check_convt_ln(h, Z);
        }
        if ((rc = finish_ssrn(h))) return rc;
        return oph_fetch_mag(h, Z);
    }
    if (!Y) { h->fail("the logits fetch needs the mel frames as an argument"); return OPH_ERR_INVALID; }
    if ((rc = ensure_batched_capacity(h, B))) return rc;
    g_cur = h->stream;
    const int ldy = round_up(m.n_mels, 32);
    float* dY = nullptr; float* dZ = nullptr; float* dZl = nullptr;
    const size_t zn = (size_t)B * T * m.r * m.full_dim;
    HIPCHK(h, hipMalloc((void**)&dY, (size_t)B * T * m.n_mels * 4 + (spk ? (size_t)B * 4 : 0)));      // (+ the speaker codes behind the frames)
    int* dSpk = spk ? (int*)(dY + (size_t)B * T * m.n_mels) : nullptr;
    if (hipMalloc((void**)&dZ, zn * 4) != hipSuccess || (Z_logits && hipMalloc((void**)&dZl, zn * 4) != hipSuccess)) {
        hipFree(dY); if (dZ) hipFree(dZ); (void)hipGetLastError(); h->fail("out of device memory for the SSRN output"); return OPH_ERR_DEVICE;
    }
    hipError_t e = hipMemcpyAsync(dY, Y, (size_t)B * T * m.n_mels * 4, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess && spk) e = hipMemcpyAsync(dSpk, spk, (size_t)B * 4, hipMemcpyHostToDevice, h->stream);
    hipError_t es = hipSuccess;
    for (int attempt = 0; attempt < 2 && e == hipSuccess; ++attempt) {
        launch_pad_rows(dY, m.n_mels, h->actB, ldy, (long long)B * T, m.n_mels, h->stream);
        rc = run_ssrn_on(h, h->actB, ldy, B, T, dZ, 0, dZl, dSpk);
        e = hipMemcpyAsync(Z, dZ, zn * 4, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess && Z_logits) e = hipMemcpyAsync(Z_logits, dZl, zn * 4, hipMemcpyDeviceToHost, h->stream);
        es = hipStreamSynchronize(h->stream);
        if (rc || es != hipSuccess || !h->host_prog || h->host_prog[8] == 0) break;
        // a fused conv1d_transpose + LayerNorm launch timed out (its workgroups were not co-resident): once more with two launches per layer
        h->host_prog[8] = 0; h->pg_ln_off = true; h->n_recoveries++;
    }
    hipFree(dY); hipFree(dZ); if (dZl) hipFree(dZl);
    if (e != hipSuccess || es != hipSuccess) { h->fail("ssrn failed: %s", hipGetErrorString(e != hipSuccess ? e : es)); return OPH_ERR_DEVICE; }
    return rc;
}
int oph_ssrn(oph_handle* h, const float* Y, int B, int T, float* Z) { return ssrn_common(h, Y, B, T, Z, nullptr); }
int oph_ssrn_speakers(oph_handle* h, const float* Y, const int32_t* spk, int B, int T, float* Z, float* Z_logits) {
    if (h && (!Y || !spk)) { h->fail("null argument"); return OPH_ERR_INVALID; }
    return ssrn_common(h, Y, B, T, Z, Z_logits, spk);
}
int oph_ssrn_logits(oph_handle* h, const float* Y, int B, int T, float* Z, float* Z_logits) {
    if (h && !Z_logits) { h->fail("null argument"); return OPH_ERR_INVALID; }
    return ssrn_common(h, Y, B, T, Z, Z_logits);
}

// ---- measurement ----------------------------------------------------------------------------
int oph_timer_start(oph_handle* h) {
    if (!h) return OPH_ERR_INVALID;
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    return OPH_OK;
}
int oph_timer_stop(oph_handle* h, float* ms) {
    if (!h || !ms) return OPH_ERR_INVALID;
    if (h->pipelined && h->sssrn) {      // join the pipelined SSRN stream first
        HIPCHK(h, hipEventRecord(h->ev_dec_done, h->sssrn));
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_dec_done, 0));
    }
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    HIPCHK(h, hipEventSynchronize(h->ev1));
    HIPCHK(h, hipEventElapsedTime(ms, h->ev0, h->ev1));
    return OPH_OK;
}
// Device-side witness of the whole-decode launches since the last reset: every workgroup stores the constant 100 MHz clock when it
// enters (minimum kept) and when it leaves (maximum kept); *total_us = sum over launches of (last out - first in).
int oph_loop_clock(oph_handle* h, int64_t* launches, double* total_us, int reset) {
    if (!h) return OPH_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    if (h->sdec) HIPCHK(h, hipStreamSynchronize(h->sdec));
    drain_loop_clock(h);
    if (launches) *launches = h->clk_launches;
    if (total_us) *total_us = h->clk_total_us;
    if (reset) { h->clk_launches = 0; h->clk_total_us = 0; }
    return OPH_OK;
}
int oph_profile_enable(oph_handle* h, int on) { if (!h) return OPH_ERR_INVALID; h->profiling = on == 2 ? 2 : (on != 0); return OPH_OK; }
int oph_profile_reset(oph_handle* h) {
    if (!h) return OPH_ERR_INVALID;
    hipStreamSynchronize(h->stream);
    for (auto& pc : h->prof) { pc.launches = 0; pc.bytes = pc.flops = pc.ms = 0; pc.used = 0; }
    return OPH_OK;
}
int oph_profile_count(const oph_handle* h) { return h ? PC_COUNT : OPH_ERR_INVALID; }
int oph_profile_get(oph_handle* h, int index, char* name, int name_cap, int64_t* launches, double* total_ms,
                    double* alg_bytes, double* alg_flops) {
    if (!h || index < 0 || index >= PC_COUNT) return OPH_ERR_INVALID;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipStreamSynchronize(h->scone));
    HIPCHK(h, hipStreamSynchronize(h->sdec));
    if (h->sssrn) HIPCHK(h, hipStreamSynchronize(h->sssrn));
    ProfClass& pc = h->prof[index];
    double ms = 0;
    for (size_t i = 0; i < pc.used; ++i) {
        float t = 0;
        if (hipEventElapsedTime(&t, pc.ev[i].first, pc.ev[i].second) == hipSuccess) ms += t;
    }
    pc.ms = ms;
    if (name && name_cap > 0) { strncpy(name, pc.name, name_cap - 1); name[name_cap - 1] = 0; }
    if (launches) *launches = pc.launches;
    if (total_ms) *total_ms = ms;
    if (alg_bytes) *alg_bytes = pc.bytes;
    if (alg_flops) *alg_flops = pc.flops;
    return OPH_OK;
}

}  // extern "C"
