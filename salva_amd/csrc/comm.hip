// comm.hip — the two Transport implementations (comm.h).
#include "comm.h"

#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <vector>

#include "common.h"

namespace salva {

// ---------------------------------------------------------------------------------------------------
// Loopback: N worlds in one process, one host thread each.  A generation-counting barrier separates "everyone has
// published its send pointers" from "everyone has finished copying".
// ---------------------------------------------------------------------------------------------------
struct LoopbackShared {
    int size;
    std::mutex mu;
    std::condition_variable cv;
    int waiting = 0;
    uint64_t generation = 0;
    struct Box {
        const void* send_lo = nullptr; size_t n_lo = 0;
        const void* send_hi = nullptr; size_t n_hi = 0;
        uint64_t cnt_to_lo[2] = {0, 0}, cnt_to_hi[2] = {0, 0};
        double red_f64[64];
        unsigned long long red_u64[64];
        const unsigned long long* gather_src = nullptr;  // allgather_u64: where this rank's section lies (device memory of the one process)
    };
    std::vector<Box> box;
    explicit LoopbackShared(int n) : size(n), box(n) {}
    void barrier() {
        std::unique_lock<std::mutex> lk(mu);
        const uint64_t gen = generation;
        if (++waiting == size) {
            waiting = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return generation != gen; });
        }
    }
};

std::shared_ptr<LoopbackShared> loopback_create(int size) { return std::make_shared<LoopbackShared>(size); }

class LoopbackTransport : public Transport {
  public:
    LoopbackTransport(std::shared_ptr<LoopbackShared> g, int r) : g_(std::move(g)), rank_(r) {}
    int rank() const override { return rank_; }
    int size() const override { return g_->size; }

    static bool trace() { static const bool t = getenv("SALVA_HIP_DIST_TRACE") != nullptr; return t; }
    void sendrecv(const void* send_lo, size_t n_lo, const void* send_hi, size_t n_hi, void* recv_lo, size_t m_lo, void* recv_hi,
                  size_t m_hi, hipStream_t s) override {
        if (trace()) fprintf(stderr, "[loopback %d] sendrecv send %zu/%zu recv %zu/%zu\n", rank_, n_lo, n_hi, m_lo, m_hi);
        SALVA_HIP_CHECK(hipStreamSynchronize(s));  // my send buffers are complete
        auto& me = g_->box[rank_];
        me.send_lo = send_lo; me.n_lo = n_lo; me.send_hi = send_hi; me.n_hi = n_hi;
        g_->barrier();
        if (has_lo() && m_lo) {
            const auto& nb = g_->box[rank_ - 1];
            if (nb.n_hi != m_lo) throw HipError(-1, "loopback sendrecv: size mismatch with the lower neighbour");
            SALVA_HIP_CHECK(hipMemcpyAsync(recv_lo, nb.send_hi, m_lo, hipMemcpyDeviceToDevice, s));
        }
        if (has_hi() && m_hi) {
            const auto& nb = g_->box[rank_ + 1];
            if (nb.n_lo != m_hi) throw HipError(-1, "loopback sendrecv: size mismatch with the upper neighbour");
            SALVA_HIP_CHECK(hipMemcpyAsync(recv_hi, nb.send_lo, m_hi, hipMemcpyDeviceToDevice, s));
        }
        SALVA_HIP_CHECK(hipStreamSynchronize(s));
        g_->barrier();  // neighbours may now reuse their send buffers
    }

    void exchange_counts(const uint64_t to_lo[2], const uint64_t to_hi[2], uint64_t from_lo[2], uint64_t from_hi[2],
                         hipStream_t) override {
        if (trace()) fprintf(stderr, "[loopback %d] exchange_counts %llu %llu\n", rank_, (unsigned long long)to_lo[0], (unsigned long long)to_hi[0]);
        auto& me = g_->box[rank_];
        memcpy(me.cnt_to_lo, to_lo, sizeof(me.cnt_to_lo));
        memcpy(me.cnt_to_hi, to_hi, sizeof(me.cnt_to_hi));
        g_->barrier();
        from_lo[0] = from_lo[1] = from_hi[0] = from_hi[1] = 0;
        if (has_lo()) memcpy(from_lo, g_->box[rank_ - 1].cnt_to_hi, 2 * sizeof(uint64_t));
        if (has_hi()) memcpy(from_hi, g_->box[rank_ + 1].cnt_to_lo, 2 * sizeof(uint64_t));
        g_->barrier();
    }

    void allreduce_sum_f32(float* buf, int n, hipStream_t s) override {
        for (; n > 64; n -= 64, buf += 64) allreduce_sum_f32(buf, 64, s);  // longer vectors go piece by piece
        if (trace()) fprintf(stderr, "[loopback %d] allreduce_f32 %d\n", rank_, n);
        float h[64];
        SALVA_HIP_CHECK(hipMemcpyAsync(h, buf, n * sizeof(float), hipMemcpyDeviceToHost, s));
        SALVA_HIP_CHECK(hipStreamSynchronize(s));
        auto& me = g_->box[rank_];
        for (int k = 0; k < n; ++k) me.red_f64[k] = h[k];
        g_->barrier();
        for (int k = 0; k < n; ++k) {  // every rank adds in rank order: identical result everywhere
            float acc = 0.0f;
            for (int r = 0; r < g_->size; ++r) acc += (float)g_->box[r].red_f64[k];
            h[k] = acc;
        }
        g_->barrier();
        SALVA_HIP_CHECK(hipMemcpyAsync(buf, h, n * sizeof(float), hipMemcpyHostToDevice, s));
        SALVA_HIP_CHECK(hipStreamSynchronize(s));
    }
    void allreduce_sum_u64(unsigned long long* buf, int n, hipStream_t s) override {
        for (; n > 64; n -= 64, buf += 64) allreduce_sum_u64(buf, 64, s);
        if (trace()) fprintf(stderr, "[loopback %d] allreduce_u64 %d\n", rank_, n);
        unsigned long long h[64];
        SALVA_HIP_CHECK(hipMemcpyAsync(h, buf, n * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        SALVA_HIP_CHECK(hipStreamSynchronize(s));
        auto& me = g_->box[rank_];
        for (int k = 0; k < n; ++k) me.red_u64[k] = h[k];
        g_->barrier();
        for (int k = 0; k < n; ++k) {
            unsigned long long acc = 0;
            for (int r = 0; r < g_->size; ++r) acc += g_->box[r].red_u64[k];
            h[k] = acc;
        }
        g_->barrier();
        SALVA_HIP_CHECK(hipMemcpyAsync(buf, h, n * sizeof(unsigned long long), hipMemcpyHostToDevice, s));
        SALVA_HIP_CHECK(hipStreamSynchronize(s));
    }
    void allgather_u64(const unsigned long long* mine, unsigned long long* all, int n_each, hipStream_t s) override {
        if (n_each <= 0) return;
        if (trace()) fprintf(stderr, "[loopback %d] allgather_u64 %d\n", rank_, n_each);
        // (one address space: every rank posts where its section lies and copies the others' device to device)
        SALVA_HIP_CHECK(hipStreamSynchronize(s));
        g_->box[rank_].gather_src = mine;
        g_->barrier();
        for (int r = 0; r < g_->size; ++r)
            SALVA_HIP_CHECK(hipMemcpyAsync(all + (size_t)r * n_each, g_->box[r].gather_src, (size_t)n_each * sizeof(unsigned long long), hipMemcpyDeviceToDevice, s));
        SALVA_HIP_CHECK(hipStreamSynchronize(s));
        g_->barrier();  // (nobody reuses its section before everybody has copied it)
    }

  private:
    std::shared_ptr<LoopbackShared> g_;
    int rank_;
};

Transport* loopback_transport(const std::shared_ptr<LoopbackShared>& group, int rank) {
    if (!group || rank < 0 || rank >= group->size) throw HipError(-2, "loopback transport: bad rank");
    return new LoopbackTransport(group, rank);
}

// ---------------------------------------------------------------------------------------------------
// RCCL over xGMI: point-to-point with the two slab neighbours, grouped so both directions progress together.
// ---------------------------------------------------------------------------------------------------
#define SALVA_NCCL_CHECK(expr)                                                                        \
    do {                                                                                              \
        ncclResult_t _r = (expr);                                                                     \
        if (_r != ncclSuccess) {                                                                      \
            char _b[256];                                                                             \
            snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(_r), __FILE__, __LINE__); \
            throw ::salva::HipError(-1, _b);                                                          \
        }                                                                                             \
    } while (0)

void rccl_unique_id(unsigned char out[RCCL_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) == RCCL_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    SALVA_NCCL_CHECK(ncclGetUniqueId(&id));
    memcpy(out, &id, RCCL_ID_BYTES);
}

class RcclTransport : public Transport {
  public:
    RcclTransport(int rank, int size, const unsigned char idb[RCCL_ID_BYTES], int device) : rank_(rank), size_(size), device_(device) {
        SALVA_HIP_CHECK(hipSetDevice(device));
        ncclUniqueId id;
        memcpy(&id, idb, RCCL_ID_BYTES);
        SALVA_NCCL_CHECK(ncclCommInitRank(&comm_, size, id, rank));
        SALVA_HIP_CHECK(hipMalloc((void**)&d_cnt_, 8 * sizeof(uint64_t)));
        SALVA_HIP_CHECK(hipHostMalloc((void**)&h_cnt_, 8 * sizeof(uint64_t), hipHostMallocDefault));
    }
    ~RcclTransport() override {
        if (comm_) (void)ncclCommDestroy(comm_);
        if (d_cnt_) (void)hipFree(d_cnt_);
        if (h_cnt_) (void)hipHostFree(h_cnt_);
    }
    int rank() const override { return rank_; }
    int size() const override { return size_; }
    int device() const override { return device_; }

    void sendrecv(const void* send_lo, size_t n_lo, const void* send_hi, size_t n_hi, void* recv_lo, size_t m_lo, void* recv_hi,
                  size_t m_hi, hipStream_t s) override {
        SALVA_NCCL_CHECK(ncclGroupStart());
        if (has_lo()) {
            if (n_lo) SALVA_NCCL_CHECK(ncclSend(send_lo, n_lo, ncclChar, rank_ - 1, comm_, s));
            if (m_lo) SALVA_NCCL_CHECK(ncclRecv(recv_lo, m_lo, ncclChar, rank_ - 1, comm_, s));
        }
        if (has_hi()) {
            if (n_hi) SALVA_NCCL_CHECK(ncclSend(send_hi, n_hi, ncclChar, rank_ + 1, comm_, s));
            if (m_hi) SALVA_NCCL_CHECK(ncclRecv(recv_hi, m_hi, ncclChar, rank_ + 1, comm_, s));
        }
        SALVA_NCCL_CHECK(ncclGroupEnd());
    }

    void exchange_counts(const uint64_t to_lo[2], const uint64_t to_hi[2], uint64_t from_lo[2], uint64_t from_hi[2],
                         hipStream_t s) override {
        // d_cnt_: [0..1] to lo, [2..3] to hi, [4..5] from lo, [6..7] from hi
        h_cnt_[0] = to_lo[0]; h_cnt_[1] = to_lo[1]; h_cnt_[2] = to_hi[0]; h_cnt_[3] = to_hi[1];
        h_cnt_[4] = h_cnt_[5] = h_cnt_[6] = h_cnt_[7] = 0;
        SALVA_HIP_CHECK(hipMemcpyAsync(d_cnt_, h_cnt_, 8 * sizeof(uint64_t), hipMemcpyHostToDevice, s));
        SALVA_NCCL_CHECK(ncclGroupStart());
        if (has_lo()) {
            SALVA_NCCL_CHECK(ncclSend(d_cnt_ + 0, 2, ncclUint64, rank_ - 1, comm_, s));
            SALVA_NCCL_CHECK(ncclRecv(d_cnt_ + 4, 2, ncclUint64, rank_ - 1, comm_, s));
        }
        if (has_hi()) {
            SALVA_NCCL_CHECK(ncclSend(d_cnt_ + 2, 2, ncclUint64, rank_ + 1, comm_, s));
            SALVA_NCCL_CHECK(ncclRecv(d_cnt_ + 6, 2, ncclUint64, rank_ + 1, comm_, s));
        }
        SALVA_NCCL_CHECK(ncclGroupEnd());
        SALVA_HIP_CHECK(hipMemcpyAsync(h_cnt_, d_cnt_, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        SALVA_HIP_CHECK(hipStreamSynchronize(s));
        from_lo[0] = h_cnt_[4]; from_lo[1] = h_cnt_[5]; from_hi[0] = h_cnt_[6]; from_hi[1] = h_cnt_[7];
    }

    void allreduce_sum_f32(float* buf, int n, hipStream_t s) override {
        SALVA_NCCL_CHECK(ncclAllReduce(buf, buf, n, ncclFloat, ncclSum, comm_, s));
    }
    void allreduce_sum_u64(unsigned long long* buf, int n, hipStream_t s) override {
        SALVA_NCCL_CHECK(ncclAllReduce(buf, buf, n, ncclUint64, ncclSum, comm_, s));
    }
    void allgather_u64(const unsigned long long* mine, unsigned long long* all, int n_each, hipStream_t s) override {
        if (n_each > 0) SALVA_NCCL_CHECK(ncclAllGather(mine, all, (size_t)n_each, ncclUint64, comm_, s));
    }

  private:
    int rank_, size_, device_;
    ncclComm_t comm_ = nullptr;
    uint64_t* d_cnt_ = nullptr;
    uint64_t* h_cnt_ = nullptr;
};

// ---------------------------------------------------------------------------------------------------
// Self-test of a transport, collective: patterned messages of varying (odd, zero, longer-than-a-slot) lengths both ways,
// both all-reduces over more values than one pass holds, the count exchange.  Throws with a description of the first
// mismatch.  What tests/test_peer_transport_gpu.py and a deployment's "is the fabric wired as I think" check call.
// ---------------------------------------------------------------------------------------------------
namespace {
inline unsigned char selftest_byte(int rank, int towards_hi, int round, size_t i) {
    return (unsigned char)(((size_t)rank * 131u + (size_t)towards_hi * 17u + (size_t)round * 29u + i * 7u + (i >> 8)) & 0xffu);
}
}  // namespace

void transport_selftest(Transport& t, size_t max_bytes, int rounds, hipStream_t s) {
    const int rank = t.rank(), size = t.size();
    if (rounds < 1 || max_bytes < 16) throw HipError(-2, "transport self-test: rounds >= 1, max_bytes >= 16");
    DevBuf<unsigned char> sl, sh, rl, rh;
    sl.ensure(max_bytes); sh.ensure(max_bytes); rl.ensure(max_bytes); rh.ensure(max_bytes);
    std::vector<unsigned char> h(max_bytes), g(max_bytes);
    auto fail = [&](const char* what, int round, size_t i) {
        char b[256];
        snprintf(b, sizeof(b), "transport self-test, rank %d of %d: %s (round %d, index %zu)", rank, size, what, round, i);
        throw HipError(-1, b);
    };
    for (int r = 0; r < rounds; ++r) {
        // message length depends on the direction, the link and the round, and is 0 now and then: the link between ranks a and
        // a+1 carries len(a, 1, r) upwards and len(a+1, 0, r) downwards
        auto len = [&](int from, int towards_hi) -> size_t {
            const size_t k = (size_t)(from * 7 + towards_hi * 3 + r * 5);
            if (k % 6 == 5) return 0;
            const size_t base = max_bytes * (size_t)(r + 1) / (size_t)rounds;
            return base > (k % 13) ? base - (k % 13) : base;
        };
        const size_t n_lo = t.has_lo() ? len(rank, 0) : 0, n_hi = t.has_hi() ? len(rank, 1) : 0;
        const size_t m_lo = t.has_lo() ? len(rank - 1, 1) : 0, m_hi = t.has_hi() ? len(rank + 1, 0) : 0;
        for (size_t i = 0; i < n_lo; ++i) h[i] = selftest_byte(rank, 0, r, i);
        SALVA_HIP_CHECK(hipMemcpyAsync(sl.p, h.data(), n_lo, hipMemcpyHostToDevice, s));
        SALVA_HIP_CHECK(hipStreamSynchronize(s));
        for (size_t i = 0; i < n_hi; ++i) h[i] = selftest_byte(rank, 1, r, i);
        SALVA_HIP_CHECK(hipMemcpyAsync(sh.p, h.data(), n_hi, hipMemcpyHostToDevice, s));
        SALVA_HIP_CHECK(hipMemsetAsync(rl.p, 0xEE, max_bytes, s));
        SALVA_HIP_CHECK(hipMemsetAsync(rh.p, 0xEE, max_bytes, s));
        t.sendrecv(sl.p, n_lo, sh.p, n_hi, rl.p, m_lo, rh.p, m_hi, s);
        SALVA_HIP_CHECK(hipMemcpyAsync(g.data(), rl.p, max_bytes, hipMemcpyDeviceToHost, s));
        SALVA_HIP_CHECK(hipStreamSynchronize(s));
        for (size_t i = 0; i < m_lo; ++i)
            if (g[i] != selftest_byte(rank - 1, 1, r, i)) fail("wrong byte from the lower neighbour", r, i);
        for (size_t i = m_lo; i < max_bytes; ++i)
            if (g[i] != 0xEE) fail("bytes written beyond the message from the lower neighbour", r, i);
        SALVA_HIP_CHECK(hipMemcpyAsync(g.data(), rh.p, max_bytes, hipMemcpyDeviceToHost, s));
        SALVA_HIP_CHECK(hipStreamSynchronize(s));
        for (size_t i = 0; i < m_hi; ++i)
            if (g[i] != selftest_byte(rank + 1, 0, r, i)) fail("wrong byte from the upper neighbour", r, i);
        for (size_t i = m_hi; i < max_bytes; ++i)
            if (g[i] != 0xEE) fail("bytes written beyond the message from the upper neighbour", r, i);

        uint64_t to_lo[2] = {(uint64_t)rank * 10 + 1 + (uint64_t)r, 77}, to_hi[2] = {(uint64_t)rank * 10 + 2 + (uint64_t)r, 99}, from_lo[2], from_hi[2];
        t.exchange_counts(to_lo, to_hi, from_lo, from_hi, s);
        if (t.has_lo() && (from_lo[0] != (uint64_t)(rank - 1) * 10 + 2 + (uint64_t)r || from_lo[1] != 99)) fail("wrong counts from the lower neighbour", r, 0);
        if (t.has_hi() && (from_hi[0] != (uint64_t)(rank + 1) * 10 + 1 + (uint64_t)r || from_hi[1] != 77)) fail("wrong counts from the upper neighbour", r, 0);

        constexpr int NV = 600;  // more than one pass of any transport
        DevBuf<float> df; DevBuf<unsigned long long> du;
        df.ensure(NV); du.ensure(NV);
        std::vector<float> hf(NV); std::vector<unsigned long long> hu(NV);
        for (int k = 0; k < NV; ++k) { hf[k] = (float)(rank + 1) * 0.5f + (float)((k + r) % 17); hu[k] = (unsigned long long)(rank + 1) * 1000003ull + (unsigned long long)k * (unsigned long long)(r + 1); }
        SALVA_HIP_CHECK(hipMemcpyAsync(df.p, hf.data(), NV * sizeof(float), hipMemcpyHostToDevice, s));
        SALVA_HIP_CHECK(hipMemcpyAsync(du.p, hu.data(), NV * sizeof(unsigned long long), hipMemcpyHostToDevice, s));
        t.allreduce_sum_f32(df.p, NV, s);
        t.allreduce_sum_u64(du.p, NV, s);
        SALVA_HIP_CHECK(hipMemcpyAsync(hf.data(), df.p, NV * sizeof(float), hipMemcpyDeviceToHost, s));
        SALVA_HIP_CHECK(hipMemcpyAsync(hu.data(), du.p, NV * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        SALVA_HIP_CHECK(hipStreamSynchronize(s));
        for (int k = 0; k < NV; ++k) {
            // halves and small integers: every partial sum is exact in f32, whatever order a transport adds in
            float ef = 0.0f; unsigned long long eu = 0;
            for (int q = 0; q < size; ++q) { ef += (float)(q + 1) * 0.5f + (float)((k + r) % 17); eu += (unsigned long long)(q + 1) * 1000003ull + (unsigned long long)k * (unsigned long long)(r + 1); }
            if (hf[k] != ef) fail("wrong f32 all-reduce sum", r, (size_t)k);
            if (hu[k] != eu) fail("wrong u64 all-reduce sum", r, (size_t)k);
        }
        // the all-gather: lengths below, at and beyond one pass of the peer transport's mailbox
        const int ne = (r % 3 == 0) ? 5 : (r % 3 == 1 ? 256 : 701);
        DevBuf<unsigned long long> gm, ga;
        gm.ensure((size_t)ne); ga.ensure((size_t)ne * size);
        std::vector<unsigned long long> hm((size_t)ne), ha((size_t)ne * size);
        for (int k = 0; k < ne; ++k) hm[k] = ((unsigned long long)(rank + 1) << 40) ^ ((unsigned long long)k * 2654435761ull) ^ (unsigned long long)r;
        SALVA_HIP_CHECK(hipMemcpyAsync(gm.p, hm.data(), hm.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, s));
        t.allgather_u64(gm.p, ga.p, ne, s);
        SALVA_HIP_CHECK(hipMemcpyAsync(ha.data(), ga.p, ha.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        SALVA_HIP_CHECK(hipStreamSynchronize(s));
        for (int q = 0; q < size; ++q)
            for (int k = 0; k < ne; ++k)
                if (ha[(size_t)q * ne + k] != (((unsigned long long)(q + 1) << 40) ^ ((unsigned long long)k * 2654435761ull) ^ (unsigned long long)r))
                    fail("wrong all-gather entry", r, (size_t)q * ne + k);
    }
}

// Latency of the two operations a solver iteration of a decomposed run performs, collective: `iters` back-to-back exchanges
// of `bytes` bytes each way with both neighbours, then `iters` back-to-back all-reduces of four floats, each batch between
// two stream synchronisations; host wall clock / iters, in microseconds.
void transport_time(Transport& t, size_t bytes, int iters, float* us_sendrecv, float* us_allreduce, hipStream_t s) {
    if (iters < 1 || bytes < 4) throw HipError(-2, "transport timing: iters >= 1, bytes >= 4");
    DevBuf<unsigned char> sl, sh, rl, rh;
    sl.ensure(bytes); sh.ensure(bytes); rl.ensure(bytes); rh.ensure(bytes);
    DevBuf<float> red;
    red.ensure(4);
    SALVA_HIP_CHECK(hipMemsetAsync(sl.p, 1, bytes, s));
    SALVA_HIP_CHECK(hipMemsetAsync(sh.p, 2, bytes, s));
    SALVA_HIP_CHECK(hipMemsetAsync(red.p, 0, 4 * sizeof(float), s));
    const size_t lo = t.has_lo() ? bytes : 0, hi = t.has_hi() ? bytes : 0;
    auto timed = [&](auto&& op) -> float {
        for (int k = 0; k < 3; ++k) op();  // warm: first launches, and the ranks meet
        SALVA_HIP_CHECK(hipStreamSynchronize(s));
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < iters; ++k) op();
        SALVA_HIP_CHECK(hipStreamSynchronize(s));
        return (float)(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters);
    };
    const float a = timed([&] { t.sendrecv(sl.p, lo, sh.p, hi, rl.p, lo, rh.p, hi, s); });
    const float b = timed([&] { t.allreduce_sum_f32(red.p, 4, s); });
    if (us_sendrecv) *us_sendrecv = a;
    if (us_allreduce) *us_allreduce = b;
}

Transport* rccl_transport(int rank, int size, const unsigned char id[RCCL_ID_BYTES], int device) {
    if (size < 1 || rank < 0 || rank >= size) throw HipError(-2, "rccl transport: bad rank / size");
    return new RcclTransport(rank, size, id, device);
}

}  // namespace salva
