// A small process-wide worker pool for the host-side segment emission (token-level timestamps): the window sums of a chunk's
// tokens are sequential f32 sums — 70-250 us on one core per 30 s chunk — and independent of each other, so they are spread
// over a few threads that stay alive between calls (spawning threads per call cost as much as the work).
// run(n, fn) executes fn(0..n-1) on the workers and the caller and returns when all are done; calls serialise.  A task that
// calls run() itself executes its tasks inline.

#include "wmi.h"

#include <atomic>
#include <condition_variable>
#include <functional>
#include <thread>

namespace wmi {

namespace {

struct Pool {
    std::vector<std::thread> th;
    std::mutex m, run_mu;
    std::condition_variable cv;
    uint64_t gen = 0;                          // generation of the current job (guarded by m)
    bool stop = false;
    const std::function<void(int)> * fn = nullptr;
    StateInstall * installs = nullptr;         // the job's caller's view of ctx.state (wmi.h: StateSlot), taken on by the workers for the job
    int n = 0;
    std::atomic<uint64_t> next{0};             // (generation << 32) | next task index: a worker of an older job can never claim
    std::atomic<int> done{0};

    explicit Pool(int workers) {
        for (int i = 0; i < workers; ++i) th.emplace_back([this] { work(); });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> lk(m); stop = true; }
        cv.notify_all();
        for (auto & t : th) t.join();
    }
    // claim and run tasks of job `g`
    void drain(uint64_t g, const std::function<void(int)> * f, int cnt) {
        for (;;) {
            uint64_t cur = next.load(std::memory_order_acquire);
            if ((cur >> 32) != (g & 0xffffffffu) || (int) (cur & 0xffffffffu) >= cnt) return;
            if (!next.compare_exchange_weak(cur, cur + 1, std::memory_order_acq_rel)) continue;
            (*f)((int) (cur & 0xffffffffu));
            done.fetch_add(1, std::memory_order_release);
        }
    }
    void work() {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int)> * f; int cnt; uint64_t g; StateInstall * inst;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return stop || gen != seen; });
                if (stop) return;
                seen = g = gen; f = fn; cnt = n; inst = installs;
            }
            t_in_task_set(true);
            state_installs_set(inst);
            drain(g, f, cnt);
            state_installs_set(nullptr);
            t_in_task_set(false);
        }
    }
    static void t_in_task_set(bool v);
};

thread_local bool t_in_task = false;
void Pool::t_in_task_set(bool v) { t_in_task = v; }

Pool * the_pool() {
    static const int workers = [] {
        unsigned hw = std::thread::hardware_concurrency();
        int w = hw > 1 ? (int) std::min(hw - 1, 7u) : 0;
        if (getenv("WMI_POOL_THREADS")) w = std::max(0, atoi(getenv("WMI_POOL_THREADS")));
        return w;
    }();
    static Pool p(workers);                    // created on first use; stopped and joined when the library is unloaded
    return &p;
}

} // namespace

void pool_run(int n_tasks, const std::function<void(int)> & fn) {
    if (n_tasks <= 0) return;
    Pool * p = the_pool();
    if (n_tasks == 1 || t_in_task || p->th.empty()) { for (int i = 0; i < n_tasks; ++i) fn(i); return; }
    std::lock_guard<std::mutex> run_lk(p->run_mu);
    uint64_t g;
    {
        std::lock_guard<std::mutex> lk(p->m);
        g = ++p->gen; p->fn = &fn; p->n = n_tasks; p->installs = state_installs_top();
        p->done.store(0, std::memory_order_relaxed);
        p->next.store((g & 0xffffffffu) << 32, std::memory_order_release);
    }
    p->cv.notify_all();
    auto wrapped_drain = [&] { t_in_task = true; p->drain(g, &fn, n_tasks); t_in_task = false; };
    wrapped_drain();                           // the caller works too
    while (p->done.load(std::memory_order_acquire) < n_tasks) __builtin_ia32_pause();
}

bool pool_in_task() { return t_in_task; }

} // namespace wmi
