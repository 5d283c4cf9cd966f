// world_diag.hip — kernel-development diagnostics of World (SALVA_HIP_DIAG builds only: `make VARIANT=diag`).  Not part of
// libsalva_hip.so; bench.py and the tests never load the diag library.
#include <algorithm>
#include <cstring>
#include <vector>

#include "../kernels.h"
#include "../world.h"

namespace salva {

// per-tile phase cycle counts of k_pred_density (SALVA_HIP_TILE_TIMING=1)
void World::tile_timing_report() {
    {
        DevBuf<unsigned long long> dbg;
        const size_t nt = std::max<uint32_t>(last_ctx.nlaunch, 1u);
        dbg.ensure(nt * 8);
        SALVA_HIP_CHECK(hipMemsetAsync(dbg.p, 0, nt * 8 * sizeof(unsigned long long), stream));
        StepCtx cd = last_ctx;
        cd.dbg = dbg.p;
        cd.ctl = nullptr;
        launch_pred_density(cd, lds, last_dt, stream);
        std::vector<unsigned long long> h(nt * 8);
        SALVA_HIP_CHECK(hipMemcpyAsync(h.data(), dbg.p, nt * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
        SALVA_HIP_CHECK(hipStreamSynchronize(stream));
        double ph[5] = {0, 0, 0, 0, 0}; size_t cnt = 0; unsigned long long tmin = ~0ull, tmax = 0;
        for (size_t t = 0; t < nt; ++t) {
            const unsigned long long* d = &h[t * 8];
            if (d[5] == 0) continue;
            for (int k = 0; k < 5; ++k) ph[k] += (double)(d[k + 1] - d[k]);
            tmin = std::min(tmin, d[0]); tmax = std::max(tmax, d[5]);
            ++cnt;
        }
        fprintf(stderr, "[tile timing] %zu tiles: setup %.0f | stage issue %.0f | barrier wait %.0f | compute %.0f | finish %.0f cycles (avg per tile); kernel span %.0f cycles\n",
                cnt, ph[0] / cnt, ph[1] / cnt, ph[2] / cnt, ph[3] / cnt, ph[4] / cnt, (double)(tmax - tmin));
    }
}

// Diagnostics: one variant of k_pred_density on the last step's state; the checksum of the kappa it wrote lets the
// caller verify that all variants compute the same bits.
float World::time_variant(int variant, uint32_t param, int reps, uint64_t* checksum) {
    use_device();
    if (!have_last_ctx || !sorted_valid || n == 0) throw HipError(SALVA_HIP_E_INVALID, "no completed step to time");
    if (reps < 1) reps = 1;
    if (variant == 2 && !pipe.fits(2, 0, 2, true)) throw HipError(SALVA_HIP_E_CAPACITY, "the pipeline does not fit the LDS for this scene");
    if (const char* e = getenv("SALVA_HIP_PIPE_WAVES")) pipe.threads = (uint32_t)std::min<int>(std::max(atoi(e), 1), PIPE_MAX_WAVES) * WAVE;
    DevBuf<uint32_t> arrivals;
    arrivals.ensure(std::max<uint32_t>(4096u, last_ctx.nlaunch));
    SALVA_HIP_CHECK(hipMemsetAsync(kappa.p, 0xff, (size_t)n * sizeof(float), stream));
    SALVA_HIP_CHECK(hipMemsetAsync(partials.p, 0xff, (size_t)last_ctx.nlaunch * last_ctx.nmodels * sizeof(float), stream));
    if (getenv("SALVA_HIP_TILE_TIMING") && (variant == 2 || variant == 4)) {
        // per-wave phase stamps of every tile: where a tile's time goes in the persistent kernels
        DevBuf<unsigned long long> dbg;
        const size_t nrec = (size_t)std::max<uint32_t>(last_ctx.nlaunch, 1u) * PIPE_MAX_WAVES;
        dbg.ensure(nrec * 8);
        SALVA_HIP_CHECK(hipMemsetAsync(dbg.p, 0, nrec * 8 * sizeof(unsigned long long), stream));
        StepCtx cd = last_ctx;
        cd.dbg = dbg.p;
        launch_pred_density_variant(cd, lds, pipe, last_dt, variant, param, arrivals.p, stream);
        std::vector<unsigned long long> h(nrec * 8);
        SALVA_HIP_CHECK(hipMemcpyAsync(h.data(), dbg.p, nrec * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
        SALVA_HIP_CHECK(hipStreamSynchronize(stream));
        double ph[5] = {0, 0, 0, 0, 0}, ph_max[5] = {0, 0, 0, 0, 0}; size_t cnt = 0; unsigned long long tmin = ~0ull, tmax = 0;
        for (size_t r = 0; r < nrec; ++r) {
            const unsigned long long* d = &h[r * 8];
            if (d[5] == 0) continue;
            for (int q = 0; q < 5; ++q) { const double v = (double)(d[q + 1] - d[q]); ph[q] += v; ph_max[q] = std::max(ph_max[q], v); }
            tmin = std::min(tmin, d[0]); tmax = std::max(tmax, d[5]);
            ++cnt;
        }
        if (cnt) {
            if (variant == 4)
                fprintf(stderr, "[variant 4 timing] %zu wave-tiles: top wait+barrier A %.0f | issue %.0f | vm wait %.0f | barrier B %.0f | compute %.0f cycles (avg per wave); kernel span %.0f\n",
                        cnt, ph[0] / cnt, ph[1] / cnt, ph[2] / cnt, ph[3] / cnt, ph[4] / cnt, (double)(tmax - tmin));
            else
                fprintf(stderr, "[variant 2 timing] %zu wave-tiles: top vm wait %.0f | barrier %.0f | issue+prefetch %.0f | - | compute %.0f cycles (avg per wave); kernel span %.0f\n",
                        cnt, ph[0] / cnt, ph[1] / cnt, ph[2] / cnt, ph[4] / cnt, (double)(tmax - tmin));
        }
    }
    if (variant == 8) {
        // dispatch order for the experiment "heaviest tiles first": block b works on order[b]; within the blocks of one XCD
        // (b % 8) the slots of that XCD's contiguous eighth, by descending particle count (stable: neighbours stay together)
        const uint32_t nl = last_ctx.nlaunch;
        std::vector<uint4> info(nl);
        SALVA_HIP_CHECK(hipMemcpy(info.data(), slot_info.p, (size_t)nl * sizeof(uint4), hipMemcpyDeviceToHost));
        std::vector<uint32_t> order(std::max<uint32_t>(nl, 4096u));
        const uint32_t q = nl >> 3, r = nl & 7u;
        for (uint32_t x = 0; x < 8; ++x) {
            const uint32_t base = (x < r) ? x * (q + 1u) : r * (q + 1u) + (x - r) * q, len = q + (x < r ? 1u : 0u);
            std::vector<uint32_t> sl(len);
            for (uint32_t k = 0; k < len; ++k) sl[k] = base + k;
            const int by = (int)param;
            std::stable_sort(sl.begin(), sl.end(), [&](uint32_t a, uint32_t b) {
                const uint32_t ca = by == 1 ? (info[a].w & 0xffffu) : info[a].y - info[a].x, cb = by == 1 ? (info[b].w & 0xffffu) : info[b].y - info[b].x;
                return ca > cb;
            });
            for (uint32_t k = 0; k < len; ++k) order[x + 8u * k] = sl[k];
        }
        SALVA_HIP_CHECK(hipMemcpy(arrivals.p, order.data(), (size_t)nl * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    launch_pred_density_variant(last_ctx, lds, pipe, last_dt, variant, param, arrivals.p, stream);  // warm-up + checksum run
    std::vector<uint32_t> hk((size_t)n + (size_t)last_ctx.nlaunch * last_ctx.nmodels);
    SALVA_HIP_CHECK(hipMemcpyAsync(hk.data(), kappa.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, stream));
    SALVA_HIP_CHECK(hipMemcpyAsync(hk.data() + n, partials.p, (size_t)last_ctx.nlaunch * last_ctx.nmodels * sizeof(float), hipMemcpyDeviceToHost, stream));
    SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    if (checksum) {
        uint64_t hsh = 1469598103934665603ull;
        // kappa only: the per-tile error partials depend (in their last bits) on how slices map to waves
        for (size_t q = 0; q < (size_t)n; ++q) { hsh ^= hk[q]; hsh *= 1099511628211ull; }
        *checksum = hsh;
        if (getenv("SALVA_HIP_VARIANT_VERBOSE")) {
            double tot = 0.0;
            for (size_t q = n; q < hk.size(); ++q) { float f; memcpy(&f, &hk[q], 4); tot += f; }
            fprintf(stderr, "[variant %d] sum of error partials %.9g\n", variant, tot);
        }
    }
    SALVA_HIP_CHECK(hipEventRecord(ev[0], stream));
    for (int r = 0; r < reps; ++r) launch_pred_density_variant(last_ctx, lds, pipe, last_dt, variant, param, arrivals.p, stream);
    SALVA_HIP_CHECK(hipEventRecord(ev[1], stream));
    SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    float ms = 0.0f;
    SALVA_HIP_CHECK(hipEventElapsedTime(&ms, ev[0], ev[1]));
    return ms * 1000.0f / (float)reps;
}

}  // namespace salva
