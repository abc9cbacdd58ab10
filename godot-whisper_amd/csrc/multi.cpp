// One host process, several GPUs (a Godot host is one process).  The reference's precedent is
// whisper_full_parallel (W/whisper.cpp:5837-5913): one shared read-only model, one whisper_state and one worker thread per
// piece, results gathered by the caller.  Here a worker is a GPU:
//   * device 0 parses the model file and builds the weight arena; every other device gets the ~1 MB header image and a
//     peer-to-peer copy of the arena over xGMI (the in-process form of "one broadcast of the packed device arena": nothing
//     is re-parsed, re-converted or re-quantised per GPU);
//   * chunk c goes to device c mod N; each device's chunks advance in lock-step on that device (wmi_full_batch), one host
//     thread per device; no collective, no cross-device traffic after the load;
//   * results stay in the owning context; wmi_pool_select() routes the whisper_full_get_* accessors to it.

#include "wmi.h"

#include <thread>

using namespace wmi;

struct wmi_pool {
    std::vector<whisper_context *> ctx;
    std::vector<int> devices;
    std::vector<int> owner, index;          // per chunk of the last call: owning context, position in its batch
    std::vector<int64_t> t_us;              // per device: wall time of its share of the last call
};

extern "C" {

void wmi_pool_free(struct wmi_pool * p) {
    if (!p) return;
    for (whisper_context * c : p->ctx) whisper_free(c);
    delete p;
}

struct wmi_pool * wmi_pool_init(const void * model, size_t model_size, const int * devices, int n_devices) {
    if (!model || !devices || n_devices < 1 || n_devices > 64) return nullptr;
    wmi_pool * p = nullptr;
    try {
        p = new wmi_pool();
        p->devices.assign(devices, devices + n_devices);
        whisper_context * c0 = init_context(model, model_size, devices[0], true);
        if (!c0) { delete p; return nullptr; }
        p->ctx.push_back(c0);
        const std::vector<uint8_t> header = export_header(c0->model, (const uint8_t *) model, c0->w.arena_bytes);
        for (int d = 1; d < n_devices; ++d) {
            whisper_context * c = init_context(header.data(), header.size(), devices[d], true, true);     // arena laid out and zeroed, weights pending
            if (!c || c->w.arena_bytes != c0->w.arena_bytes) { WMI_ERR("%s: device %d: arena layout mismatch\n", __func__, devices[d]); if (c) whisper_free(c); wmi_pool_free(p); return nullptr; }
            p->ctx.push_back(c);
            bool ok;
            if (devices[d] == devices[0]) ok = HIP_OK(hipMemcpy(c->w.arena, c0->w.arena, c0->w.arena_bytes, hipMemcpyDeviceToDevice));
            else {
                int can = 0; (void) hipDeviceCanAccessPeer(&can, devices[d], devices[0]);
                ok = HIP_OK(hipMemcpyPeer(c->w.arena, devices[d], c0->w.arena, devices[0], c0->w.arena_bytes));     // staged through the host if P2P is off
                (void) can;
            }
            if (!ok || !HIP_OK(hipDeviceSynchronize())) { wmi_pool_free(p); return nullptr; }
            c->weights_pending = false;                     // the peer copy has landed
        }
    } catch (const std::exception & e) {
        WMI_ERR("%s: %s\n", __func__, e.what());
        if (p) wmi_pool_free(p);
        return nullptr;
    }
    p->t_us.assign(n_devices, 0);
    return p;
}

int wmi_pool_size(struct wmi_pool * p) { return p ? (int) p->ctx.size() : 0; }
struct whisper_context * wmi_pool_context(struct wmi_pool * p, int i) { return (p && i >= 0 && i < (int) p->ctx.size()) ? p->ctx[i] : nullptr; }

int wmi_pool_full(struct wmi_pool * p, struct whisper_full_params params, const float * const * pcm, const int * n_samples, int n_chunks) {
    if (!p || !pcm || !n_samples || n_chunks < 0) return -1;
    const int N = (int) p->ctx.size();
    try {
    p->owner.assign(n_chunks, 0); p->index.assign(n_chunks, 0);
    std::vector<std::vector<const float *>> ptrs(N); std::vector<std::vector<int>> lens(N);
    for (int c = 0; c < n_chunks; ++c) {
        const int d = c % N;
        p->owner[c] = d; p->index[c] = (int) ptrs[d].size();
        ptrs[d].push_back(pcm[c]); lens[d].push_back(n_samples[c]);
    }
    params.no_context = true;
    std::vector<int> rets(N, 0);
    // every entry point of a context serialises on its mutex (wmi_device.h): the workers take it like any other caller, so a
    // concurrent wmi_batch_select / whisper_full on one of the pool's contexts waits instead of racing; nothing may throw out of a
    // worker thread (std::terminate) or across the C boundary
    auto work = [&](int d) {
        const int64_t t0 = time_us();
        try {
            std::lock_guard<std::recursive_mutex> lk(p->ctx[d]->mu);
            if (!ptrs[d].empty()) {
                (void) hipSetDevice(p->ctx[d]->device);
                rets[d] = full_batch(*p->ctx[d], params, ptrs[d].data(), lens[d].data(), (int) ptrs[d].size(), false);
            } else if (p->ctx[d]->batch) { p->ctx[d]->batch->results.clear(); p->ctx[d]->batch->redo.clear(); }
        } catch (const std::exception & e) {
            WMI_ERR("wmi_pool_full: device %d: %s\n", p->ctx[d]->device, e.what());
            rets[d] = -9;
        } catch (...) { rets[d] = -9; }
        p->t_us[d] = time_us() - t0;
    };
    {
        std::vector<std::thread> th;
        struct Joiner { std::vector<std::thread> & t; ~Joiner() { for (auto & x : t) if (x.joinable()) x.join(); } } joiner{th};
        for (int d = 1; d < N; ++d) th.emplace_back(work, d);
        work(0);
    }
    for (int d = 0; d < N; ++d) if (rets[d] != 0) return rets[d];
    return 0;
    } catch (const std::exception & e) {                    // allocation of the work lists / thread creation
        WMI_ERR("wmi_pool_full: %s\n", e.what());
        return -9;
    }
}

struct whisper_context * wmi_pool_select(struct wmi_pool * p, int chunk) {
    if (!p || chunk < 0 || chunk >= (int) p->owner.size()) return nullptr;
    whisper_context * c = p->ctx[p->owner[chunk]];
    return wmi_batch_select(c, p->index[chunk]) >= 0 ? c : nullptr;
}

int64_t wmi_pool_device_time_us(struct wmi_pool * p, int i) { return (p && i >= 0 && i < (int) p->t_us.size()) ? p->t_us[i] : -1; }

} // extern "C"
