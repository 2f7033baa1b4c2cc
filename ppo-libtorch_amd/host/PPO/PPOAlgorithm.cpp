#include "PPOAlgorithm.h"

#include <cmath>
#include <filesystem>
#include <fstream>
#include <thread>

#include "../Config/Toml.h"
#include "../Utils/TorchArchive.h"
#include "PPO_Discrete.h"
#include "PPO_MultiDiscrete.h"

using ppo::DType;
using ppo::Tensor;
namespace fs = std::filesystem;

PPOAlgorithm::PPOAlgorithm(int env_kind, int dist_kind, int64_t default_obs, int64_t default_max_episode_steps)
    : m_env_kind(env_kind), m_dist_kind(dist_kind) {
    // defaults: PPO_Discrete.cpp:7-29 (MultiDiscrete: max_episode_steps 200, PPO_MultiDiscrete.cpp:31)
    m_obs_size = default_obs;
    m_action_size = 1;
    m_action_high = 1.0f;
    m_action_low = -1.0f;
    m_learning_rate = 0.0003f;
    m_seed = 1;
    m_total_timesteps = 100000;
    m_use_cuda = true;
    m_torch_deterministic = true;
    m_num_envs = 1;
    m_num_steps = 2048;
    m_anneal_lr = false;
    m_use_gae = true;
    m_gamma = 0.99f;
    m_gae_lambda = 0.95f;
    m_num_minibatches = 64;
    m_update_epochs = 10;
    m_norm_adv = true;
    m_clip_coef = 0.2f;
    m_clip_vloss = true;
    m_ent_coef = 0.01f;
    m_vf_coef = 0.5f;
    m_max_grad_norm = 0.5f;
    m_checkpoint_updates = 5;
    m_max_episode_steps = default_max_episode_steps;
    m_batch_size = m_num_envs * m_num_steps;
    m_minibatch_size = m_batch_size / m_num_minibatches;
    m_global_step = 0;
}

PPOAlgorithm::~PPOAlgorithm() {
    m_agent.reset();
    if (m_ctx) ppo_ctx_destroy(m_ctx);
}

void PPOAlgorithm::getArgs() {
    const std::string path = "./PPOConfig.toml";
    if (!ppo::FlatToml::exists(path)) {
        std::cout << "Could not find " << path << " file." << "\nUsing default PPO hyperparameters" << std::endl;   // the reference's words (:112-113)
        return;
    }
    try {
        ppo::FlatToml cfg(path);
        auto I = [&](const char* sec, const char* key, int64_t& dst) { if (auto v = cfg.integer(sec, key)) { dst = *v; std::cout << "Using config file " << key << " = " << dst << std::endl; } };
        auto F = [&](const char* sec, const char* key, float& dst) { if (auto v = cfg.real(sec, key)) { dst = *v; std::cout << "Using config file " << key << " = " << dst << std::endl; } };
        auto B = [&](const char* sec, const char* key, bool& dst) { if (auto v = cfg.boolean(sec, key)) { dst = *v; std::cout << "Using config file " << key << " = " << (dst ? "true" : "false") << std::endl; } };
        I("environment", "obs_size", m_obs_size);
        I("environment", "action_size", m_action_size);
        if (m_env_kind == PPO_ENV_MOUNTAINCAR) {   // PPO_MultiDiscrete only, between action_size and max_episode_steps (PPO_MultiDiscrete.cpp:136-144)
            F("environment", "action_high", m_action_high);
            F("environment", "action_low", m_action_low);
        }
        I("environment", "max_episode_steps", m_max_episode_steps);
        I("general", "seed", m_seed);
        I("general", "total_timesteps", m_total_timesteps);
        B("general", "use_cuda", m_use_cuda);
        B("general", "torch_deterministic", m_torch_deterministic);
        I("general", "checkpoint_updates", m_checkpoint_updates);
        F("ppo", "learning_rate", m_learning_rate);
        I("ppo", "num_envs", m_num_envs);
        I("ppo", "num_steps", m_num_steps);
        B("ppo", "anneal_lr", m_anneal_lr);
        B("ppo", "use_gae", m_use_gae);
        F("ppo", "gamma", m_gamma);
        F("ppo", "gae_lambda", m_gae_lambda);
        I("ppo", "num_minibatches", m_num_minibatches);
        I("ppo", "update_epochs", m_update_epochs);
        B("ppo", "norm_adv", m_norm_adv);
        F("ppo", "clip_coef", m_clip_coef);
        B("ppo", "clip_vloss", m_clip_vloss);
        F("ppo", "ent_coef", m_ent_coef);
        F("ppo", "vf_coef", m_vf_coef);
        F("ppo", "max_grad_norm", m_max_grad_norm);
        m_batch_size = m_num_envs * m_num_steps;                 // :246-247
        m_minibatch_size = m_batch_size / m_num_minibatches;
    } catch (const ppo::FlatToml::ParseError& err) {
        std::cerr << "Error parsing config file: " << err.what() << "\nat " << path << ":" << err.line << "\n"
                  << "Using default PPO hyperparameters" << std::endl;
    }
}

Tensor PPOAlgorithm::bufferView(int which, std::vector<int64_t> shape, DType dt) const {
    void* p = nullptr;
    size_t bytes = 0;
    ppo::check(ppo_buffer(m_ctx, which, &p, &bytes), m_ctx, "ppo_buffer");
    return Tensor::view(m_device, p, std::move(shape), dt);
}

void PPOAlgorithm::construct() {
    m_threadPool = std::make_shared<ThreadPool>(static_cast<int64_t>(std::thread::hardware_concurrency()));
    if (!m_use_cuda) {
        // The reference honours use_cuda = false by running on the CPU (PPO_Discrete.cpp:65), and its shipped CartPoleRecommendedSettings.toml
        // sets it.  This build has no CPU path: the same configuration trains, on the GPU, and says so.
        std::cout << "Warning: use_cuda = false requested, but this build has no CPU path; training on the gfx950 (HIP) device." << std::endl;
    }
    m_device = std::make_shared<ppo::Device>(0);
    std::cout << "Using gfx950 (HIP) device " << m_device->ordinal() << std::endl;
    std::cout << "m_obs_size: " << m_obs_size << std::endl;
    std::cout << "m_action_size: " << m_action_size << std::endl;

    ppo_config c{};
    c.struct_size = sizeof c;
    c.device = m_device->ordinal();
    c.env_kind = m_env_kind;
    c.dist_kind = m_dist_kind;
    c.obs_size = static_cast<int32_t>(m_obs_size);
    c.n_heads = 1;
    c.head_dims[0] = static_cast<int32_t>(m_action_size);      // m_actionSpace = { actionSize }, Agent.cpp:21
    c.hidden = 64; c.n_hidden = 2;
    c.num_envs = static_cast<int32_t>(m_num_envs);
    c.num_steps = static_cast<int32_t>(m_num_steps);
    c.num_minibatches = static_cast<int32_t>(m_num_minibatches);
    c.update_epochs = static_cast<int32_t>(m_update_epochs);
    c.max_episode_steps = static_cast<int32_t>(m_max_episode_steps);
    c.use_gae = m_use_gae; c.norm_adv = m_norm_adv; c.clip_vloss = m_clip_vloss; c.anneal_lr = m_anneal_lr;
    c.seed = m_seed;
    c.total_timesteps = m_total_timesteps;
    c.learning_rate = m_learning_rate; c.gamma = m_gamma; c.gae_lambda = m_gae_lambda; c.clip_coef = m_clip_coef;
    c.ent_coef = m_ent_coef; c.vf_coef = m_vf_coef; c.max_grad_norm = m_max_grad_norm;
    ppo::check(ppo_ctx_create(&c, &m_ctx), nullptr, "PPO");    // obs-size mismatch surfaces here with the reference's message (:370-375)
    ppo::check(ppo_params_init_orthogonal(m_ctx, m_seed), m_ctx, "agent init");
    m_agent = std::make_shared<Agent>(m_ctx, m_device, std::vector<int64_t>{ m_action_size });

    loadPolicyFromCheckpoint();
    std::cout << "made envs" << std::endl;

    const int64_t T = m_num_steps, N = m_num_envs;
    m_obs = bufferView(PPO_BUF_OBS, { T, N, m_obs_size }, DType::f32);
    m_actions = bufferView(PPO_BUF_ACTIONS, { T, N, 1 }, DType::i32);
    m_logprobs = bufferView(PPO_BUF_LOGPROBS, { T, N }, DType::f32);
    m_rewards = bufferView(PPO_BUF_REWARDS, { T, N }, DType::f32);
    m_dones = bufferView(PPO_BUF_DONES, { T, N }, DType::f32);
    m_values = bufferView(PPO_BUF_VALUES, { T, N }, DType::f32);
    if (m_dist_kind == PPO_DIST_MASKED) m_action_masks = bufferView(PPO_BUF_MASKS, { T, N, m_action_size }, DType::u8);
    m_episode_stats = std::make_unique<CircularBuffer>(static_cast<size_t>(100));
}

AgentOutput PPOAlgorithm::actImpl(const Tensor& obs, const Tensor* mask, const Tensor& action) const {
    return mask ? m_agent->getActionAndValueMasked(obs, *mask, action) : m_agent->getActionAndValueDiscrete(obs, action);
}

Tensor PPOAlgorithm::initEnvsImpl() {
    ppo::check(ppo_env_reset(m_ctx), m_ctx, "initEnvs");
    ppo::check(ppo_sync(m_ctx), m_ctx, "sync");
    return bufferView(PPO_BUF_NEXT_OBS, { m_num_envs, m_obs_size }, DType::f32);
}

std::tuple<Tensor, Tensor, Tensor> PPOAlgorithm::stepEnvs(const Tensor& action) {
    const int64_t N = m_num_envs;
    Tensor obs(m_device, { N, m_obs_size }, DType::f32), reward(m_device, { N, 1 }, DType::f32), done(m_device, { N, 1 }, DType::i32);
    ppo::check(ppo_env_step(m_ctx, action.data<int64_t>(), obs.data<float>(), reward.data<float>(), done.data<int32_t>()), m_ctx, "stepEnvs");
    ppo::check(ppo_sync(m_ctx), m_ctx, "sync");
    return { obs, reward, done };
}

std::array<Tensor, 2> PPOAlgorithm::calcAdvantage(const Tensor& next_obs, const Tensor& next_done) const {
    // the reference bootstraps from the tensors it is handed; they are the context's NEXT_OBS / NEXT_DONE unless the caller made its own
    void *po = nullptr, *pd = nullptr;
    size_t bo = 0, bd = 0;
    ppo::check(ppo_buffer(m_ctx, PPO_BUF_NEXT_OBS, &po, &bo), m_ctx, "ppo_buffer");
    ppo::check(ppo_buffer(m_ctx, PPO_BUF_NEXT_DONE, &pd, &bd), m_ctx, "ppo_buffer");
    if (next_obs.defined() && next_obs.data_ptr() != po) ppo::check(ppo_memcpy_h2d(m_ctx, po, next_obs.cpu<float>().data(), bo), m_ctx, "next_obs");
    if (next_done.defined() && next_done.data_ptr() != pd) ppo::check(ppo_memcpy_h2d(m_ctx, pd, next_done.cpu<int32_t>().data(), bd), m_ctx, "next_done");
    ppo::check(ppo_calc_advantage(m_ctx), m_ctx, "calcAdvantage");
    ppo::check(ppo_sync(m_ctx), m_ctx, "sync");
    return { bufferView(PPO_BUF_RETURNS, { m_num_steps, m_num_envs }, DType::f32), bufferView(PPO_BUF_ADVANTAGES, { m_num_steps, m_num_envs }, DType::f32) };
}

// Off the hot path (the fused kernel forms these statistics itself): API-compatible helper on caller tensors.
Tensor PPOAlgorithm::getApproxKLAndClippedObj(const Tensor& ratio, const Tensor& logratio) {
    const std::vector<float> r = ratio.cpu<float>(), lr = logratio.cpu<float>();
    int64_t clipped = 0;
    double kl = 0.0;
    for (size_t i = 0; i < r.size(); i++) {
        if (std::fabs(r[i] - 1.0f) > m_clip_coef) clipped++;
        kl += static_cast<double>((r[i] - 1.0f) - lr[i]);
    }
    m_clipfracs.push_back(static_cast<float>(clipped) / static_cast<float>(r.size()));
    return Tensor::from_host<float>(m_device, { static_cast<float>(kl / static_cast<double>(r.size())) }, { 1 });
}

void PPOAlgorithm::train() {
    m_threadPool->start();
    uint64_t global_step = m_global_step;
    const auto start_time = std::chrono::steady_clock::now();
    auto update_time = start_time;
    const int64_t num_updates = (m_total_timesteps - static_cast<int64_t>(global_step)) / m_batch_size;   // :496
    ppo::check(ppo_env_reset(m_ctx), m_ctx, "initEnvs");

    // One iteration = LR anneal (:514-518), rollout (:524-548), advantages (:554), all epochs x minibatches (:567-644), explained variance (:647-648),
    // and a statistics snapshot behind them.  Everything here is ENQUEUED: nothing waits for the GPU.
    auto enqueue = [&](int64_t update) {
        if (m_anneal_lr) {   // num_updates is recomputed from the remaining steps after a resume, like the reference
            const double frac = 1.0 - (update - 1.0) / num_updates;
            ppo::check(ppo_set_learning_rate(m_ctx, frac * m_learning_rate), m_ctx, "lr");
        }
        ppo::check(ppo_rollout(m_ctx, nullptr), m_ctx, "rollout");
        ppo::check(ppo_calc_advantage(m_ctx), m_ctx, "calcAdvantage");
        ppo::check(ppo_update(m_ctx), m_ctx, "update");
        ppo::check(ppo_stats_snapshot(m_ctx), m_ctx, "stats");
    };
    if (num_updates >= 1) enqueue(1);
    for (int64_t update = 1; update < num_updates + 1; update++) {
        // The reference's loop is serial: step, print, step.  Here the GPU starts iteration update + 1 while the host waits for, formats and prints
        // the table of iteration `update` (the statistics were snapshotted behind its last kernel) -- except across a checkpoint, which must read
        // the parameters iteration `update` left.
        const bool checkpoint_due = update % m_checkpoint_updates == 0;
        if (update < num_updates && !checkpoint_due) enqueue(update + 1);
        ppo::check(ppo_stats_snapshot_read(m_ctx, &m_last_stats), m_ctx, "stats");   // waits for iteration `update` only
        const ppo_stats& st = m_last_stats;
        global_step += static_cast<uint64_t>(m_batch_size);
        m_episode_stats->assign(st.ep_len_mean, static_cast<float>(st.ep_rew_mean), static_cast<size_t>(st.ep_count));
        m_clipfracs.assign(1, static_cast<float>(st.clipfrac_mean));

        const auto end = std::chrono::steady_clock::now();
        const auto time_elapsed = std::chrono::duration_cast<std::chrono::milliseconds>(end - start_time);
        const auto fps = std::chrono::duration_cast<std::chrono::milliseconds>(end - update_time);
        auto scalar = [&](double v) { return Tensor::host_scalar(static_cast<float>(v)); };
        Tensor kl = scalar(st.approx_kl), ent = scalar(st.entropy_loss), ev = scalar(st.explained_variance), loss = scalar(st.loss),
               pg = scalar(st.pg_loss), vl = scalar(st.v_loss);
        m_last_stats_valid = true;
        printPPOResults(update, static_cast<int64_t>(global_step), fps, time_elapsed, kl, ent, ev, loss, pg, vl);
        m_last_stats_valid = false;
        update_time = std::chrono::steady_clock::now();

        if (checkpoint_due) {   // :662-673
            fs::create_directories("./ModelCheckpoints/");
            fs::create_directories("./OptimizerCheckpoints/");
            const std::string a = "./ModelCheckpoints/PPO_Agent_" + std::to_string(global_step) + "_steps.pt";
            const std::string o = "./OptimizerCheckpoints/PPO_Optimizer_" + std::to_string(global_step) + "_steps.pt";
            std::cout << "Saving model checkpoint to " << a << "..." << std::endl;
            std::cout << "Saving optimizer checkpoint to " << o << "..." << std::endl;
            saveCheckpoint(a, o);
            if (update < num_updates) enqueue(update + 1);
        }
    }
    fs::create_directories("./Models/");   // :678-685
    const std::string a = "./Models/PPO_Agent_" + std::to_string(m_total_timesteps) + "_steps.pt";
    const std::string o = "./Models/PPO_Optimizer_" + std::to_string(m_total_timesteps) + "_steps.pt";
    std::cout << "Saving model " << a << "..." << std::endl;
    std::cout << "Saving optimizer " << o << "..." << std::endl;
    saveCheckpoint(a, o);
    m_global_step = global_step;
    m_threadPool->stop();
}

// Checkpoint files keep the reference's directories, names AND container: LibTorch module archives (Utils/TorchArchive.h), so a run can
// be resumed from files the reference wrote and the reference can load what this build writes.  The agent archive holds the twelve
// tensors of Agent::parameters(), the optimizer archive AdamW's step / exp_avg / exp_avg_sq per parameter and the options it ran with.
void PPOAlgorithm::saveCheckpoint(const std::string& agentFile, const std::string& optimizerFile) {
    const int64_t P = ppo_param_count(m_ctx);
    std::vector<float> p(static_cast<size_t>(P)), m(static_cast<size_t>(P)), v(static_cast<size_t>(P));
    int64_t step = 0;
    ppo::check(ppo_params_get_h(m_ctx, p.data(), P), m_ctx, "params");
    ppo::check(ppo_optimizer_get_h(m_ctx, m.data(), v.data(), P, &step), m_ctx, "optimizer");
    ppo_stats st{};
    ppo::check(ppo_read_stats(m_ctx, &st), m_ctx, "stats");
    // written under a temporary name and renamed once complete and flushed: a full disk never leaves a truncated file where the newest-file
    // rule of loadPolicyFromCheckpoint would pick it up.  (The archive's internal directory is named after the final file, as LibTorch's is.)
    const std::string ta = agentFile + ".tmp", to = optimizerFile + ".tmp";
    try {
        ppo::pt::writeAgent(ta, m_obs_size, 64, m_action_size, p, fs::path(agentFile).stem().string());
        ppo::pt::writeOptimizer(to, m_obs_size, 64, m_action_size, m, v, step, st.learning_rate, static_cast<double>(1e-5f), 0.01,
                                     fs::path(optimizerFile).stem().string());
    } catch (const std::exception&) {
        std::error_code ec;
        fs::remove(ta, ec); fs::remove(to, ec);
        throw std::runtime_error("could not write checkpoint " + agentFile + " / " + optimizerFile);
    }
    fs::rename(ta, agentFile);
    fs::rename(to, optimizerFile);
}

static std::string newestFile(const fs::path& dir) {
    std::string best;
    fs::file_time_type when{};
    for (const auto& e : fs::directory_iterator(dir)) {
        if (e.path().extension() == ".tmp") continue;   // an interrupted save
        if (best.empty() || fs::last_write_time(e) > when) { best = e.path().string(); when = fs::last_write_time(e); }
    }
    return best;
}
// The tensors of an archive, flattened in file order, if they are exactly this agent's twelve (shape by shape); otherwise the reason is
// printed and the file ignored -- the agent then starts fresh, as when no checkpoint exists.
static bool flattenFor(const std::vector<ppo::pt::NamedTensor>& tensors, const std::vector<std::vector<int64_t>>& shapes, const std::string& file,
                       std::vector<float>& flat) {
    if (tensors.size() != shapes.size()) {
        std::cout << "Checkpoint " << file << " holds " << tensors.size() << " tensors, this agent has " << shapes.size() << "; ignoring it." << std::endl;
        return false;
    }
    flat.clear();
    for (size_t i = 0; i < shapes.size(); i++) {
        if (tensors[i].sizes != shapes[i]) {
            std::cout << "Checkpoint " << file << ": tensor " << tensors[i].name << " does not have the shape of this agent's parameter " << i
                      << " (obs_size / action_size differ?); ignoring it." << std::endl;
            return false;
        }
        flat.insert(flat.end(), tensors[i].values.begin(), tensors[i].values.end());
    }
    return true;
}

void PPOAlgorithm::loadPolicyFromCheckpoint() {
    const fs::path modelDir = "./ModelCheckpoints/", optimDir = "./OptimizerCheckpoints/";
    if (!fs::exists(modelDir) || !fs::exists(optimDir)) {
        std::cout << "No previous model checkpoint found at " << modelDir << ", initializing new agent!" << std::endl;
        return;
    }
    const int64_t P = ppo_param_count(m_ctx);
    const auto shapes = ppo::pt::agentShapes(m_obs_size, 64, m_action_size);
    const std::string a = newestFile(modelDir);
    if (a.empty()) {
        std::cout << "No previous model checkpoint found at " << modelDir << ", initializing new agent!" << std::endl;
    } else {
        std::cout << "Loading model " << a << "..." << std::endl;
        try {
            std::vector<float> p;
            if (flattenFor(ppo::pt::readAgent(a).tensors, shapes, a, p)) {
                const std::string steps = PPOUtils::getLoadFromSteps(a, "PPO_Agent_");
                m_global_step = PPOUtils::isNumber(steps) ? static_cast<uint64_t>(std::stoll(steps)) : 0;   // :809-811
                std::cout << "Continuing training from step " << m_global_step << std::endl;
                ppo::check(ppo_params_set_h(m_ctx, p.data(), P), m_ctx, "load params");
            }
        } catch (const std::exception& ex) {   // not an archive, truncated, damaged (any exception, bad_alloc / length_error included): say why and start fresh
            std::cout << ex.what() << "; ignoring it." << std::endl;
        }
    }
    const std::string o = newestFile(optimDir);
    if (o.empty()) {
        std::cout << "No previous optimizer checkpoint found at " << optimDir << ", initializing new optimizer!" << std::endl;
    } else {
        std::cout << "Loading optimizer " << o << "..." << std::endl;
        try {
            const ppo::pt::OptimizerFile f = ppo::pt::readOptimizer(o);
            std::vector<float> m, v;
            bool same_step = true;
            for (int64_t s : f.step) same_step = same_step && s == f.step[0];
            if (!same_step) {
                std::cout << "Checkpoint " << o << ": the parameters are at different step counts; ignoring it." << std::endl;
            } else if (flattenFor(f.exp_avg, shapes, o, m) && flattenFor(f.exp_avg_sq, shapes, o, v)) {
                ppo::check(ppo_optimizer_set_h(m_ctx, m.data(), v.data(), P, f.step[0]), m_ctx, "load optimizer");
                // torch::load(optimizer) also brings back the options the file was saved with (:834): the learning rate is taken over (train()
                // overwrites it per update when anneal_lr is set, exactly as there); the other AdamW options are fixed in this build
                ppo::check(ppo_set_learning_rate(m_ctx, f.lr), m_ctx, "lr");
                if (f.beta1 != 0.9 || f.beta2 != 0.999 || f.eps != static_cast<double>(1e-5f) || f.weight_decay != 0.01 || f.amsgrad)
                    std::cout << "Checkpoint " << o << " was saved with AdamW options other than the reference's (betas 0.9/0.999, eps 1e-5, weight_decay 0.01);"
                              << " this build keeps the reference's." << std::endl;
            }
        } catch (const std::exception& ex) {
            std::cout << ex.what() << "; ignoring it." << std::endl;
        }
    }
}

// SB3-style table, same rows, widths and precisions as the reference prints (PPO_Discrete.cpp:700-774).
void PPOAlgorithm::printPPOResults(int64_t update, int64_t global_step, std::chrono::milliseconds fps, std::chrono::milliseconds time_elapsed,
                                   Tensor& approx_kl, Tensor& entropy_loss, Tensor& explained_var, Tensor& loss, Tensor& pg_loss, Tensor& v_loss) {
    const bool first = update == 1;
    const int w = first ? 9 : 13;
    const std::string bar(first ? 33 : 42, '-');
    auto row = [&](const std::string& label) { std::cout << "|    " << std::left << std::setw(first ? 16 : 21) << label << std::right << "| "; };
    const int64_t fps_v = static_cast<int64_t>(m_batch_size / (std::max<int64_t>(fps.count(), 1) / 1000.0));
    std::cout << bar << "\n";
    if (!m_episode_stats->empty()) {
        std::cout << (first ? "| rollout/           |          |\n" : "| rollout/                |              |\n");
        if (first) std::cout << std::setprecision(1) << std::defaultfloat; else std::cout << std::setprecision(2) << std::fixed;
        row("ep_len_mean"); printElement(m_episode_stats->avgLength(), w);
        std::cout << std::setprecision(first ? 5 : 8);
        row("ep_rew_mean"); printElement(m_episode_stats->avgReward(), w);
    }
    std::cout << (first ? "| time/              |          |\n" : "| time/                   |              |\n");
    row("fps"); printElement(fps_v, w);
    row("iterations"); printElement(update, w);
    row("time_elapsed"); printElement(static_cast<int64_t>(time_elapsed.count() / 1000.0), w);
    row("total_timesteps"); printElement(global_step, w);
    if (!first) {
        ppo_stats st = m_last_stats;
        if (!m_last_stats_valid) ppo_read_stats(m_ctx, &st);
        std::cout << "| train/                  |              |\n" << std::setprecision(9);
        row("approx_kl"); printElement(approx_kl.item<float>(), w);
        row("clip_fraction"); printElement(PPOUtils::getVectorMean(m_clipfracs), w);
        row("clip_range"); printElement(m_clip_coef, w);
        std::cout << std::fixed;
        row("entropy_loss"); printElement(entropy_loss.item<float>(), w);
        row("explained_variance"); printElement(explained_var.item<float>(), w);
        std::cout << std::defaultfloat << std::setprecision(6);
        row("learning_rate"); printElement(static_cast<float>(st.learning_rate), w);
        std::cout << std::fixed << std::setprecision(9);
        row("loss"); printElement(loss.item<float>(), w);
        row("n_updates"); printElement(update * m_update_epochs, w);
        row("policy_gradient_loss"); printElement(pg_loss.item<float>(), w);
        row("value_loss"); printElement(v_loss.item<float>(), w);
    }
    std::cout << bar << std::endl << std::endl;
}

// ---------------------------------------------------------------------------------------------------------
PPO_Discrete::PPO_Discrete() : PPOAlgorithm(PPO_ENV_CARTPOLE, PPO_DIST_CATEGORICAL, 2, 500) {
    getArgs();
    construct();
}
AgentOutput PPO_Discrete::computeActionLogic(const Tensor& next_obs) const { return actImpl(next_obs, nullptr, Tensor()); }
Tensor PPO_Discrete::initEnvs() { return initEnvsImpl(); }

PPO_MultiDiscrete::PPO_MultiDiscrete() : PPOAlgorithm(PPO_ENV_MOUNTAINCAR, PPO_DIST_MASKED, 2, 200) {
    getArgs();
    construct();
}
AgentOutput PPO_MultiDiscrete::computeActionLogic(const Tensor& next_obs, const Tensor& action_mask, const Tensor& action) {
    return actImpl(next_obs, &action_mask, action);
}
Tensor PPO_MultiDiscrete::initEnvs(const Tensor& action_mask) {
    if (action_mask.defined()) {   // m_envs[i]->getActionMask(): ones (MountainCar.cpp:69-77)
        Tensor ones = action_mask;
        ones.copy_from_host(std::vector<uint8_t>(static_cast<size_t>(action_mask.numel()), 1));
    }
    return initEnvsImpl();
}
