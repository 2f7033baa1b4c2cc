// Drop-in for the reference's Utils/ThreadPool.h:14-75 (start / queueJob / waitForJobsToFinish / stop / busy).  The HIP path
// does not use it -- the reference needs it to step one env per job (PPO_Discrete.cpp:429-468), here one kernel steps them
// all -- but host code written against the reference's pool keeps compiling.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <queue>
#include <thread>
#include <vector>

class ThreadPool {
  public:
    ThreadPool() noexcept : ThreadPool(1) {}
    explicit ThreadPool(int64_t numThreads) noexcept : m_count(numThreads < 1 ? 1 : numThreads) {}
    ~ThreadPool() noexcept { if (!m_workers.empty()) stop(); }
    ThreadPool(const ThreadPool&) = delete;
    ThreadPool& operator=(const ThreadPool&) = delete;

    void start() noexcept {
        m_quit = false;
        for (int64_t i = 0; i < m_count; i++) m_workers.emplace_back([this] { work(); });
    }
    void queueJob(std::function<void()>&& job) noexcept {
        { std::lock_guard<std::mutex> g(m_mu); m_pending.push(std::move(job)); }
        m_wake.notify_one();
    }
    void waitForJobsToFinish() noexcept {
        std::unique_lock<std::mutex> g(m_mu);
        m_idle.wait(g, [this] { return m_pending.empty() && m_running == 0; });
    }
    void stop() noexcept {
        { std::lock_guard<std::mutex> g(m_mu); m_quit = true; }
        m_wake.notify_all();
        for (auto& t : m_workers) if (t.joinable()) t.join();
        m_workers.clear();
    }
    [[nodiscard]] bool busy() const noexcept { std::lock_guard<std::mutex> g(m_mu); return !m_pending.empty(); }

  private:
    void work() noexcept {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> g(m_mu);
                m_wake.wait(g, [this] { return m_quit || !m_pending.empty(); });
                if (m_pending.empty()) return;  // quitting
                job = std::move(m_pending.front());
                m_pending.pop();
                ++m_running;
            }
            try { job(); } catch (...) {}  // the reference swallows job exceptions too (ThreadPool.cpp:43-49)
            { std::lock_guard<std::mutex> g(m_mu); --m_running; }
            m_idle.notify_all();
        }
    }
    const int64_t m_count;
    mutable std::mutex m_mu;
    std::condition_variable m_wake, m_idle;
    std::queue<std::function<void()>> m_pending;
    std::vector<std::thread> m_workers;
    int m_running = 0;
    bool m_quit = false;
};
