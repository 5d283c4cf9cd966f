// comm_peer.hip — the xGMI peer-direct Transport (comm.h): neighbour exchange and the convergence all-reduce as plain stores
// into windows of device memory that every rank of the node has mapped, ordered by sequence-numbered flags.  No collective
// library call sits in the solver loop: one exchange is two short kernels on the world's stream (put: copy my bytes into the
// neighbour's window, fence, raise its flag; get: wait for my flags, copy out), an all-reduce of up to 256 values is one.
//
// Why it exists: an exchange of the decomposed solve moves 4-16 bytes per ghost particle (10^4-10^5 particles) once per solver
// iteration; as a grouped ncclSend/ncclRecv that costs a proxy round trip (~10-20 us) however small the message is, and the
// scalar all-reduce another.  MI355X nodes connect every GPU pair directly (7 xGMI links per GPU), so a rank can write into any
// other rank's memory; the latency of a flagged store is a few microseconds.  RCCL stays the default transport (comm.hip).
//
// Layout of a rank's window (allocated by that rank in its own HBM, fine-grained so that remote stores and system-scope flags
// are coherent; exported with hipIpcGetMemHandle, mapped by the others with hipIpcOpenMemHandle):
//     PeerHeader            flags written by the other ranks, the all-reduce mailboxes
//     data[2 sides][2][slot_bytes]   messages from the lower / upper neighbour, double-buffered by sequence parity
// Double buffering is enough without acknowledgements: calls are matched on both sides of a link and streams run in order, so a
// rank can be at most one message ahead of the neighbour that still reads the previous one (put(t+2) needs get(t+1), which
// needs the neighbour's put(t+1), which follows its get(t)).  The same argument covers the all-reduce mailboxes.
//
// A wait that is not satisfied within the transport's timeout (30 s by default: ranks reach their first exchange seconds
// apart; SALVA_HIP_PEER_TIMEOUT_S = seconds, read when the transport is connected — raise it under a debugger or when first-step
// allocations are slow, 0 = wait for ever) gives up, records the fact in a host-mapped word and lets the kernel finish: the host
// throws at its next call instead of leaving a kernel spinning.
//
// STATUS: EXPERIMENTAL.  Every run so far had all ranks on ONE GPU (the boxes available have one): the IPC mapping then stays
// inside one HBM, and no store has crossed xGMI.  What that leaves unexercised is the coherence of remote stores against the
// reader GPU's caches; the reads of a window after its flag are therefore system-scope / nontemporal loads (they must not be
// served from a line cached before the flag was seen), on top of the acquire + fence the memory model already asks for.
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "comm.h"
#include "common.h"

namespace salva {

namespace {

constexpr int PEER_MAX_RANKS = 64;
constexpr int PEER_RED_CHUNK = 256;                       // values per all-reduce pass
constexpr unsigned long long PEER_TICKS_PER_S = 100000000ull;  // wall_clock64(): 100 MHz
constexpr int PEER_COPY_BLOCKS = 64;                      // most blocks per direction of a put / get (16 KB each)
constexpr size_t PEER_ALIGN = 256;

struct PeerHeader {
    unsigned long long msg_flag[2];                       // [0] from the lower, [1] from the upper neighbour: messages arrived
    unsigned long long red_flag[PEER_MAX_RANKS];          // all-reduce contributions arrived, per contributing rank
    alignas(256) unsigned long long red[2][PEER_MAX_RANKS][PEER_RED_CHUNK];  // [parity][rank][value] (f32 in the low half)
};
constexpr size_t PEER_HEADER_BYTES = (sizeof(PeerHeader) + PEER_ALIGN - 1) / PEER_ALIGN * PEER_ALIGN;

inline size_t peer_window_bytes(size_t slot_bytes) { return PEER_HEADER_BYTES + 4 * slot_bytes; }

__device__ __forceinline__ unsigned char* peer_slot(void* window, int side, int parity, size_t slot_bytes) {
    return (unsigned char*)window + PEER_HEADER_BYTES + (size_t)(side * 2 + parity) * slot_bytes;
}

// `timeout_ticks` == 0: no timeout
__device__ __forceinline__ bool peer_wait(unsigned long long* flag, unsigned long long seq, unsigned int* timed_out,
                                          unsigned long long timeout_ticks) {
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
        __builtin_amdgcn_s_sleep(4);
        if (timeout_ticks && wall_clock64() - t0 > timeout_ticks) {
            __hip_atomic_store(timed_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return false;
        }
    }
    return true;
}

// bytes [0, n): 16-byte vectors where both pointers allow it, single bytes for the rest.  FROM_WINDOW: `src` is this rank's
// window, filled by a neighbour's stores — nontemporal loads, which do not take a line an earlier message left in a cache.
template <bool FROM_WINDOW>
__device__ __forceinline__ void peer_copy(unsigned char* dst, const unsigned char* src, size_t n, unsigned int part, unsigned int parts) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const unsigned int tid = part * blockDim.x + threadIdx.x, nthreads = parts * blockDim.x;
    size_t body = 0;
    if (((uintptr_t)dst & 15u) == 0 && ((uintptr_t)src & 15u) == 0) {
        body = n & ~(size_t)15;
        const u32x4* s4 = (const u32x4*)src;
        u32x4* d4 = (u32x4*)dst;
        for (size_t i = tid; i < body / 16; i += nthreads) d4[i] = FROM_WINDOW ? __builtin_nontemporal_load(s4 + i) : s4[i];
    }
    for (size_t i = body + tid; i < n; i += nthreads) dst[i] = FROM_WINDOW ? __builtin_nontemporal_load(src + i) : src[i];
}

struct PeerPut {
    const unsigned char* src[2];   // my send buffers: to the lower, to the upper neighbour
    size_t n[2];
    void* window[2];               // the neighbours' windows (null: no such neighbour / nothing to do on this link)
    int side_there[2];             // which of the neighbour's two receive sides I am: its upper (1) for my lower link, and vice versa
    unsigned long long seq[2];
    size_t slot_bytes;
    unsigned int* tickets;         // [2], device memory of this rank, zero between launches
};

// grid (PEER_COPY_BLOCKS, 2): y = link.  The last block of a link to finish raises the neighbour's flag.
__global__ void __launch_bounds__(256) k_peer_put(PeerPut a) {
    const int link = blockIdx.y;
    if (!a.window[link]) return;
    unsigned char* dst = peer_slot(a.window[link], a.side_there[link], (int)(a.seq[link] & 1ull), a.slot_bytes);
    peer_copy<false>(dst, a.src[link], a.n[link], blockIdx.x, gridDim.x);
    __threadfence_system();  // my stores have reached the neighbour's memory before the ticket is drawn
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int t = atomicAdd(&a.tickets[link], 1u);
        if (t + 1 == gridDim.x) {
            a.tickets[link] = 0;
            PeerHeader* h = (PeerHeader*)a.window[link];
            __hip_atomic_store(&h->msg_flag[a.side_there[link]], a.seq[link], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

struct PeerGet {
    unsigned char* dst[2];         // my receive buffers: from the lower, from the upper neighbour
    size_t n[2];
    void* window;                  // my own window
    bool active[2];
    unsigned long long seq[2];
    size_t slot_bytes;
    unsigned int* timed_out;       // host-mapped
    unsigned long long timeout_ticks;
};

__global__ void __launch_bounds__(256) k_peer_get(PeerGet a) {
    const int link = blockIdx.y;
    if (!a.active[link]) return;
    PeerHeader* h = (PeerHeader*)a.window;
    __shared__ int ok;
    if (threadIdx.x == 0) ok = peer_wait(&h->msg_flag[link], a.seq[link], a.timed_out, a.timeout_ticks) ? 1 : 0;
    __syncthreads();
    if (!ok) return;
    __threadfence_system();  // every wave of the block reads the window after the flag
    peer_copy<true>(a.dst[link], peer_slot(a.window, link, (int)(a.seq[link] & 1ull), a.slot_bytes), a.n[link], blockIdx.x, gridDim.x);
}

struct PeerReduce {
    void* windows[PEER_MAX_RANKS];
    int rank, size, n;
    unsigned long long seq;
    unsigned int* timed_out;
    unsigned long long timeout_ticks;
};

// One block of PEER_RED_CHUNK threads.  T = float or unsigned long long; every rank adds the contributions in rank order, so
// all ranks hold bit-identical sums (the solvers branch on them).
template <typename T>
__global__ void __launch_bounds__(PEER_RED_CHUNK) k_peer_allreduce(PeerReduce a, T* buf) {
    const int k = threadIdx.x, par = (int)(a.seq & 1ull);
    if (k < a.n) {
        unsigned long long bits = 0;
        const T v = buf[k];
        memcpy(&bits, &v, sizeof(T));
        for (int r = 0; r < a.size; ++r) ((PeerHeader*)a.windows[r])->red[par][a.rank][k] = bits;
    }
    __threadfence_system();
    __syncthreads();
    if (k < a.size)
        __hip_atomic_store(&((PeerHeader*)a.windows[k])->red_flag[a.rank], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    __shared__ int bad;
    if (k == 0) bad = 0;
    __syncthreads();
    PeerHeader* mine = (PeerHeader*)a.windows[a.rank];
    if (k < a.size && !peer_wait(&mine->red_flag[k], a.seq, a.timed_out, a.timeout_ticks)) bad = 1;
    __syncthreads();
    if (bad) return;
    __threadfence_system();
    if (k < a.n) {
        T acc = T(0);
        for (int r = 0; r < a.size; ++r) {
            const unsigned long long bits = __hip_atomic_load(&mine->red[par][r][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            T v;
            memcpy(&v, &bits, sizeof(T));
            acc += v;
        }
        buf[k] = acc;
    }
}

// The all-gather over the same mailbox: every rank stores its <= PEER_RED_CHUNK values into its row of every window, raises its flag,
// waits for everybody's, and copies the ROWS out side by side instead of adding them up: out[r * stride + k] = rank r's value k.
__global__ void __launch_bounds__(PEER_RED_CHUNK) k_peer_allgather(PeerReduce a, const unsigned long long* mine, unsigned long long* out, size_t stride) {
    const int k = threadIdx.x, par = (int)(a.seq & 1ull);
    if (k < a.n) {
        const unsigned long long bits = mine[k];
        for (int r = 0; r < a.size; ++r) ((PeerHeader*)a.windows[r])->red[par][a.rank][k] = bits;
    }
    __threadfence_system();
    __syncthreads();
    if (k < a.size)
        __hip_atomic_store(&((PeerHeader*)a.windows[k])->red_flag[a.rank], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    __shared__ int bad;
    if (k == 0) bad = 0;
    __syncthreads();
    PeerHeader* me = (PeerHeader*)a.windows[a.rank];
    if (k < a.size && !peer_wait(&me->red_flag[k], a.seq, a.timed_out, a.timeout_ticks)) bad = 1;
    __syncthreads();
    if (bad) return;
    __threadfence_system();
    if (k < a.n)
        for (int r = 0; r < a.size; ++r)
            out[(size_t)r * stride + k] = __hip_atomic_load(&me->red[par][r][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
struct PeerSetup {
    int rank = 0, size = 0, device = 0;
    size_t slot_bytes = 0;
    void* window = nullptr;
    hipIpcMemHandle_t handle;
    ~PeerSetup() {
        if (window) (void)hipFree(window);
    }
};

static_assert(sizeof(hipIpcMemHandle_t) == PEER_HANDLE_BYTES, "hipIpcMemHandle_t size");

PeerSetup* peer_begin(int rank, int size, int device, size_t slot_bytes, unsigned char handle_out[PEER_HANDLE_BYTES]) {
    if (size < 1 || size > PEER_MAX_RANKS || rank < 0 || rank >= size) throw HipError(-2, "peer transport: bad rank / size (at most 64 ranks)");
    if (slot_bytes < 4096) throw HipError(-2, "peer transport: slots of at least 4096 bytes");
    slot_bytes = (slot_bytes + PEER_ALIGN - 1) / PEER_ALIGN * PEER_ALIGN;
    auto s = std::make_unique<PeerSetup>();
    s->rank = rank; s->size = size; s->device = device; s->slot_bytes = slot_bytes;
    SALVA_HIP_CHECK(hipSetDevice(device));
    const size_t bytes = peer_window_bytes(slot_bytes);
    SALVA_HIP_CHECK(hipExtMallocWithFlags(&s->window, bytes, hipDeviceMallocFinegrained));
    SALVA_HIP_CHECK(hipMemset(s->window, 0, PEER_HEADER_BYTES));
    SALVA_HIP_CHECK(hipDeviceSynchronize());  // the flags read 0 before anybody can learn the handle
    SALVA_HIP_CHECK(hipIpcGetMemHandle(&s->handle, s->window));
    memcpy(handle_out, &s->handle, PEER_HANDLE_BYTES);
    return s.release();
}

void peer_abort(PeerSetup* s) { delete s; }

class PeerTransport : public Transport {
  public:
    // `handles`: size x PEER_HANDLE_BYTES in rank order (mine included, ignored); every rank of the node maps every window,
    // because the all-reduce writes to all of them
    PeerTransport(PeerSetup* setup, const unsigned char* handles) : rank_(setup->rank), size_(setup->size), device_(setup->device), slot_(setup->slot_bytes) {
        std::unique_ptr<PeerSetup> own(setup);
        if (const char* e = getenv("SALVA_HIP_PEER_TIMEOUT_S")) {
            char* end = nullptr;
            const double sec = strtod(e, &end);
            if (end == e || !(sec >= 0.0) || sec > 1.0e6) throw HipError(-2, "SALVA_HIP_PEER_TIMEOUT_S: seconds, 0 (no timeout) .. 1e6");
            timeout_ticks_ = (unsigned long long)(sec * (double)PEER_TICKS_PER_S);
        }
        SALVA_HIP_CHECK(hipSetDevice(device_));
        win_.assign(size_, nullptr);
        try {
            for (int r = 0; r < size_; ++r) {
                if (r == rank_) continue;
                hipIpcMemHandle_t h;
                memcpy(&h, handles + (size_t)r * PEER_HANDLE_BYTES, PEER_HANDLE_BYTES);
                SALVA_HIP_CHECK(hipIpcOpenMemHandle(&win_[r], h, hipIpcMemLazyEnablePeerAccess));
            }
            SALVA_HIP_CHECK(hipMalloc((void**)&tickets_, 2 * sizeof(unsigned int)));
            SALVA_HIP_CHECK(hipMemset(tickets_, 0, 2 * sizeof(unsigned int)));
            SALVA_HIP_CHECK(hipMalloc((void**)&d_cnt_, 8 * sizeof(uint64_t)));
            SALVA_HIP_CHECK(hipHostMalloc((void**)&h_cnt_, 8 * sizeof(uint64_t), hipHostMallocDefault));
            SALVA_HIP_CHECK(hipHostMalloc((void**)&timed_out_, sizeof(unsigned int), hipHostMallocMapped | hipHostMallocCoherent));
            *timed_out_ = 0;
            SALVA_HIP_CHECK(hipDeviceSynchronize());
        } catch (...) {
            release();
            throw;
        }
        win_[rank_] = own->window;
        own->window = nullptr;  // mine from here on
    }
    ~PeerTransport() override {
        (void)hipSetDevice(device_);
        (void)hipDeviceSynchronize();
        if (win_.size() == (size_t)size_ && win_[rank_]) {
            (void)hipFree(win_[rank_]);
            win_[rank_] = nullptr;
        }
        release();
    }
    int rank() const override { return rank_; }
    int size() const override { return size_; }
    int device() const override { return device_; }

    void sendrecv(const void* send_lo, size_t n_lo, const void* send_hi, size_t n_hi, void* recv_lo, size_t m_lo, void* recv_hi,
                  size_t m_hi, hipStream_t s) override {
        check_timeout();
        // a link carries both directions in lock-step: max(1, ceil(longer direction / slot)) rounds, the same on both ends
        const bool act[2] = {has_lo(), has_hi()};
        const unsigned char* src[2] = {(const unsigned char*)send_lo, (const unsigned char*)send_hi};
        unsigned char* dst[2] = {(unsigned char*)recv_lo, (unsigned char*)recv_hi};
        const size_t n[2] = {n_lo, n_hi}, m[2] = {m_lo, m_hi};
        size_t rounds[2] = {0, 0}, most = 0;
        for (int l = 0; l < 2; ++l) {
            if (!act[l]) continue;
            const size_t longer = n[l] > m[l] ? n[l] : m[l];
            rounds[l] = longer ? (longer + slot_ - 1) / slot_ : 1;
            most = rounds[l] > most ? rounds[l] : most;
        }
        for (size_t r = 0; r < most; ++r) {
            PeerPut put{};
            PeerGet get{};
            put.slot_bytes = get.slot_bytes = slot_;
            put.tickets = tickets_;
            get.window = win_[rank_];
            get.timed_out = timed_out_;
            get.timeout_ticks = timeout_ticks_;
            for (int l = 0; l < 2; ++l) {
                if (!act[l] || r >= rounds[l]) continue;
                const size_t off = r * slot_;
                const unsigned long long seq = ++msg_seq_[l];
                put.src[l] = src[l] ? src[l] + off : nullptr;
                put.n[l] = n[l] > off ? (n[l] - off < slot_ ? n[l] - off : slot_) : 0;
                put.window[l] = win_[l == 0 ? rank_ - 1 : rank_ + 1];
                put.side_there[l] = 1 - l;
                put.seq[l] = seq;
                get.dst[l] = dst[l] ? dst[l] + off : nullptr;
                get.n[l] = m[l] > off ? (m[l] - off < slot_ ? m[l] - off : slot_) : 0;
                get.active[l] = true;
                get.seq[l] = seq;
            }
            hipLaunchKernelGGL(k_peer_put, dim3(copy_blocks(put.n[0], put.n[1]), 2), dim3(256), 0, s, put);
            hipLaunchKernelGGL(k_peer_get, dim3(copy_blocks(get.n[0], get.n[1]), 2), dim3(256), 0, s, get);
        }
        SALVA_HIP_CHECK(hipGetLastError());
    }

    void exchange_counts(const uint64_t to_lo[2], const uint64_t to_hi[2], uint64_t from_lo[2], uint64_t from_hi[2],
                         hipStream_t s) override {
        // d_cnt_: [0..1] to lo, [2..3] to hi, [4..5] from lo, [6..7] from hi
        h_cnt_[0] = to_lo[0]; h_cnt_[1] = to_lo[1]; h_cnt_[2] = to_hi[0]; h_cnt_[3] = to_hi[1];
        h_cnt_[4] = h_cnt_[5] = h_cnt_[6] = h_cnt_[7] = 0;
        SALVA_HIP_CHECK(hipMemcpyAsync(d_cnt_, h_cnt_, 8 * sizeof(uint64_t), hipMemcpyHostToDevice, s));
        sendrecv(d_cnt_ + 0, 2 * sizeof(uint64_t), d_cnt_ + 2, 2 * sizeof(uint64_t), d_cnt_ + 4, 2 * sizeof(uint64_t), d_cnt_ + 6,
                 2 * sizeof(uint64_t), s);
        SALVA_HIP_CHECK(hipMemcpyAsync(h_cnt_, d_cnt_, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        SALVA_HIP_CHECK(hipStreamSynchronize(s));
        check_timeout();
        from_lo[0] = h_cnt_[4]; from_lo[1] = h_cnt_[5]; from_hi[0] = h_cnt_[6]; from_hi[1] = h_cnt_[7];
    }

    void allreduce_sum_f32(float* buf, int n, hipStream_t s) override { allreduce(buf, n, s); }
    void allreduce_sum_u64(unsigned long long* buf, int n, hipStream_t s) override { allreduce(buf, n, s); }
    void allgather_u64(const unsigned long long* mine, unsigned long long* all, int n_each, hipStream_t s) override {
        check_timeout();
        for (int at = 0; at < n_each; at += PEER_RED_CHUNK) {
            PeerReduce a{};
            for (int r = 0; r < size_; ++r) a.windows[r] = win_[r];
            a.rank = rank_; a.size = size_; a.n = n_each - at < PEER_RED_CHUNK ? n_each - at : PEER_RED_CHUNK;
            a.seq = ++red_seq_;
            a.timed_out = timed_out_;
            a.timeout_ticks = timeout_ticks_;
            hipLaunchKernelGGL(k_peer_allgather, dim3(1), dim3(PEER_RED_CHUNK), 0, s, a, mine + at, all + at, (size_t)n_each);
        }
        SALVA_HIP_CHECK(hipGetLastError());
    }

  private:
    static unsigned int copy_blocks(size_t a, size_t b) {
        const size_t longer = a > b ? a : b, blocks = (longer + 16383) / 16384;
        return (unsigned int)(blocks < 1 ? 1 : blocks > (size_t)PEER_COPY_BLOCKS ? (size_t)PEER_COPY_BLOCKS : blocks);
    }
    template <typename T>
    void allreduce(T* buf, int n, hipStream_t s) {
        check_timeout();
        for (; n > 0; n -= PEER_RED_CHUNK, buf += PEER_RED_CHUNK) {
            PeerReduce a{};
            for (int r = 0; r < size_; ++r) a.windows[r] = win_[r];
            a.rank = rank_; a.size = size_; a.n = n < PEER_RED_CHUNK ? n : PEER_RED_CHUNK;
            a.seq = ++red_seq_;
            a.timed_out = timed_out_;
            a.timeout_ticks = timeout_ticks_;
            hipLaunchKernelGGL(k_peer_allreduce<T>, dim3(1), dim3(PEER_RED_CHUNK), 0, s, a, buf);
        }
        SALVA_HIP_CHECK(hipGetLastError());
    }
    void check_timeout() {
        if (timed_out_ && *(volatile unsigned int*)timed_out_)
            throw HipError(-1, "peer transport: a neighbour's message did not arrive within " + std::to_string(timeout_ticks_ / PEER_TICKS_PER_S) +
                                   " s (a rank died, or the ranks' calls do not match; SALVA_HIP_PEER_TIMEOUT_S moves the limit)");
    }
    void release() {
        for (int r = 0; r < (int)win_.size(); ++r)
            if (r != rank_ && win_[r]) (void)hipIpcCloseMemHandle(win_[r]);
        win_.clear();
        if (tickets_) (void)hipFree(tickets_);
        if (d_cnt_) (void)hipFree(d_cnt_);
        if (h_cnt_) (void)hipHostFree(h_cnt_);
        if (timed_out_) (void)hipHostFree(timed_out_);
        tickets_ = nullptr; d_cnt_ = nullptr; h_cnt_ = nullptr; timed_out_ = nullptr;
    }

    int rank_, size_, device_;
    size_t slot_;
    std::vector<void*> win_;
    unsigned long long msg_seq_[2] = {0, 0}, red_seq_ = 0;
    unsigned long long timeout_ticks_ = 30ull * PEER_TICKS_PER_S;
    unsigned int* tickets_ = nullptr;
    uint64_t* d_cnt_ = nullptr;
    uint64_t* h_cnt_ = nullptr;
    unsigned int* timed_out_ = nullptr;
};

Transport* peer_transport(PeerSetup* setup, const unsigned char* handles) {
    if (!setup || !handles) throw HipError(-2, "peer transport: null argument");
    return new PeerTransport(setup, handles);
}

}  // namespace salva
