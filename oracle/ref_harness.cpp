// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Drives the *unmodified* reference (AidanShipperley/PPO-LibTorch) where it lies under
// /root/reference, linked against the LibTorch that ships inside the torch wheel of this image.
// It is built only by oracle/Makefile into oracle/_ref/ (git-ignored) and is used to
//   (1) generate the golden vectors committed under tests/golden/   (mode "golden")
//   (2) time the reference's own CPU ThreadPool path                 (mode "bench")
// Nothing under ppo-libtorch_amd/ links, includes or executes this file.
//
// All members of the reference's algorithm classes are public (PPO/PPO_Discrete.h:24-108), so the
// harness calls initEnvs/stepEnvs/computeActionLogic/calcAdvantage one by one and reads the m_*
// buffers.  PPO_Discrete::train() is monolithic, so its minibatch loop (PPO_Discrete.cpp:567-644)
// is re-driven here with the same LibTorch calls in the same RNG-consuming order; the harness then
// CERTIFIES that re-drive by running the real train() on a twin instance and demanding
// bit-identical final parameters ("certified_bitwise" entry in every scenario).
//
// Golden container ("PGLD1"): u32 count, then per entry {u32 name_len, name, u32 dtype, u32 ndim,
// i64 dims[ndim], raw little-endian data}.  dtype: 0=f32 1=i64 2=i32 3=u8 4=f64.

#include "PPO/PPO_Discrete.h"
#include "PPO/PPO_MultiDiscrete.h"

#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <random>
#include <sstream>
#include <unistd.h>

namespace {

struct Entry {
    std::string name;
    uint32_t dtype;
    std::vector<int64_t> dims;
    std::vector<char> bytes;
};

struct GoldWriter {
    std::vector<Entry> entries;

    void addRaw(const std::string& name, uint32_t dtype, std::vector<int64_t> dims, const void* p, size_t nbytes) {
        Entry e;
        e.name = name;
        e.dtype = dtype;
        e.dims = std::move(dims);
        e.bytes.assign(static_cast<const char*>(p), static_cast<const char*>(p) + nbytes);
        entries.push_back(std::move(e));
    }
    void add(const std::string& name, const torch::Tensor& tin) {
        torch::Tensor t = tin.detach().cpu().contiguous();
        uint32_t dt;
        if (t.scalar_type() == torch::kFloat32) dt = 0;
        else if (t.scalar_type() == torch::kInt64) dt = 1;
        else if (t.scalar_type() == torch::kInt32) dt = 2;
        else if (t.scalar_type() == torch::kBool) { t = t.to(torch::kUInt8); dt = 3; }
        else if (t.scalar_type() == torch::kUInt8) dt = 3;
        else if (t.scalar_type() == torch::kFloat64) dt = 4;
        else throw std::runtime_error("GoldWriter: unsupported dtype for " + name);
        addRaw(name, dt, t.sizes().vec(), t.data_ptr(), static_cast<size_t>(t.nbytes()));
    }
    void addF32(const std::string& name, const std::vector<float>& v, std::vector<int64_t> dims) {
        addRaw(name, 0, std::move(dims), v.data(), v.size() * sizeof(float));
    }
    void addI64(const std::string& name, const std::vector<int64_t>& v, std::vector<int64_t> dims) {
        addRaw(name, 1, std::move(dims), v.data(), v.size() * sizeof(int64_t));
    }
    void addF64(const std::string& name, const std::vector<double>& v, std::vector<int64_t> dims) {
        addRaw(name, 4, std::move(dims), v.data(), v.size() * sizeof(double));
    }
    void save(const std::string& path) const {
        std::ofstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error("cannot open " + path);
        const char magic[8] = { 'P', 'G', 'L', 'D', '1', 0, 0, 0 };
        f.write(magic, 8);
        uint32_t n = static_cast<uint32_t>(entries.size());
        f.write(reinterpret_cast<const char*>(&n), 4);
        for (const Entry& e : entries) {
            uint32_t nl = static_cast<uint32_t>(e.name.size());
            f.write(reinterpret_cast<const char*>(&nl), 4);
            f.write(e.name.data(), nl);
            f.write(reinterpret_cast<const char*>(&e.dtype), 4);
            uint32_t nd = static_cast<uint32_t>(e.dims.size());
            f.write(reinterpret_cast<const char*>(&nd), 4);
            f.write(reinterpret_cast<const char*>(e.dims.data()), nd * sizeof(int64_t));
            f.write(e.bytes.data(), static_cast<std::streamsize>(e.bytes.size()));
        }
        std::cerr << "[ref_harness] wrote " << path << " (" << entries.size() << " entries)\n";
    }
};

struct RunCfg {
    int64_t obs_size = 4, action_size = 2, max_episode_steps = 500;
    int64_t seed = 2, total_timesteps = 0;  // 0 => updates * batch
    int64_t num_envs = 8, num_steps = 32, num_minibatches = 4, update_epochs = 10;
    bool anneal_lr = true, use_gae = true, norm_adv = true, clip_vloss = true;
    double learning_rate = 0.001, gamma = 0.98, gae_lambda = 0.95, clip_coef = 0.2, ent_coef = 0.0, vf_coef = 0.5,
           max_grad_norm = 0.5;
    int64_t updates = 1;
};

std::string makeScratchDir(const std::string& tag) {
    std::string tmpl = "/tmp/ppo_ref_" + tag + "_XXXXXX";
    std::vector<char> buf(tmpl.begin(), tmpl.end());
    buf.push_back(0);
    if (!mkdtemp(buf.data())) throw std::runtime_error("mkdtemp failed");
    return std::string(buf.data());
}

// The reference reads ./PPOConfig.toml from the CWD (PPO_Discrete.cpp:108).
void enterScratchWithConfig(const RunCfg& c, const std::string& tag) {
    std::string dir = makeScratchDir(tag);
    if (chdir(dir.c_str()) != 0) throw std::runtime_error("chdir failed");
    std::ofstream f("PPOConfig.toml");
    auto b = [](bool v) { return v ? "true" : "false"; };
    int64_t total = c.total_timesteps ? c.total_timesteps : c.updates * c.num_envs * c.num_steps;
    f << std::setprecision(17);
    f << "[environment]\nobs_size = " << c.obs_size << "\naction_size = " << c.action_size
      << "\nmax_episode_steps = " << c.max_episode_steps << "\n\n";
    f << "[general]\nseed = " << c.seed << "\ntotal_timesteps = " << total
      << "\nuse_cuda = false\ntorch_deterministic = true\ncheckpoint_updates = 1000000\n\n";
    f << "[ppo]\nlearning_rate = " << c.learning_rate << "\nnum_envs = " << c.num_envs << "\nnum_steps = " << c.num_steps
      << "\nanneal_lr = " << b(c.anneal_lr) << "\nuse_gae = " << b(c.use_gae) << "\ngamma = " << c.gamma
      << "\ngae_lambda = " << c.gae_lambda << "\nnum_minibatches = " << c.num_minibatches
      << "\nupdate_epochs = " << c.update_epochs << "\nnorm_adv = " << b(c.norm_adv) << "\nclip_coef = " << c.clip_coef
      << "\nclip_vloss = " << b(c.clip_vloss) << "\nent_coef = " << c.ent_coef << "\nvf_coef = " << c.vf_coef
      << "\nmax_grad_norm = " << c.max_grad_norm << "\n";
}

torch::Tensor flatParams(const std::vector<torch::Tensor>& ps, bool grads = false) {
    std::vector<torch::Tensor> parts;
    for (const auto& p : ps) parts.push_back((grads ? p.grad() : p).detach().reshape(-1));
    return torch::cat(parts).clone();
}

template <class Optim>
std::array<torch::Tensor, 2> flatAdamState(Optim& opt, const std::vector<torch::Tensor>& ps) {
    std::vector<torch::Tensor> m, v;
    for (const auto& p : ps) {
        auto& st = static_cast<torch::optim::AdamWParamState&>(*opt.state().at(p.unsafeGetTensorImpl()));
        m.push_back(st.exp_avg().detach().reshape(-1));
        v.push_back(st.exp_avg_sq().detach().reshape(-1));
    }
    return { torch::cat(m).clone(), torch::cat(v).clone() };
}

// One minibatch of train()'s update loop up to and including backward() -- the expressions of PPO_Discrete.cpp:576-638 /
// PPO_MultiDiscrete.cpp:602-665 in their order, through the reference's own Agent / distributions / getApproxKLAndClippedObj and LibTorch autograd.
// Every re-drive of this file (goldTrainScenario, goldHeadline, goldConfig4) goes through THIS function, and goldTrainScenario certifies it
// bit-identical to the reference's real train() ("certified_bitwise").  The caller clips and steps (:640-641), so it can record in between.
struct MinibatchOut {
    AgentOutput o;
    torch::Tensor pg_loss, v_loss, entropy_loss, approx_kl, loss;
};
template <class Algo, bool Masked>
MinibatchOut minibatchLossBackward(Algo& algo, const torch::Tensor& mb, const torch::Tensor& b_obs, const torch::Tensor& b_masks, const torch::Tensor& b_actions,
                                   const torch::Tensor& b_logprobs, const torch::Tensor& b_advantages, const torch::Tensor& b_returns, const torch::Tensor& b_values) {
    MinibatchOut r;
    if constexpr (Masked)
        r.o = algo.m_agent->getActionAndValueMasked(b_obs.index({ mb }), b_masks.index({ mb }), b_actions.to(torch::kLong).index({ mb }).t());
    else
        r.o = algo.m_agent->getActionAndValueDiscrete(b_obs.index({ mb }), b_actions.to(torch::kLong).index({ mb }));
    torch::Tensor logratio = r.o.logprob - b_logprobs.index({ mb });
    torch::Tensor ratio = logratio.exp();
    r.approx_kl = algo.getApproxKLAndClippedObj(ratio, logratio);
    torch::Tensor adv = b_advantages.index({ mb });
    if (algo.m_norm_adv) adv = (adv - adv.mean()) / (adv.std() + 1e-8f);
    torch::Tensor l1 = -adv * ratio;
    torch::Tensor l2 = -adv * torch::clamp(ratio, 1 - algo.m_clip_coef, 1 + algo.m_clip_coef);
    r.pg_loss = torch::max(l1, l2).mean();
    torch::Tensor nv = r.o.value.view(-1);
    torch::Tensor ret = b_returns.index({ mb });
    if (algo.m_clip_vloss) {
        torch::Tensor un = (nv - ret) * (nv - ret);
        torch::Tensor vold = b_values.index({ mb });
        torch::Tensor vc = vold + torch::clamp(nv - vold, -algo.m_clip_coef, algo.m_clip_coef);
        torch::Tensor cl = (vc - ret) * (vc - ret);
        r.v_loss = 0.5f * torch::max(un, cl).mean();
    } else {
        r.v_loss = 0.5f * ((nv - ret) * (nv - ret)).mean();
    }
    r.entropy_loss = r.o.entropy.mean();
    r.loss = r.pg_loss - algo.m_ent_coef * r.entropy_loss + r.v_loss * algo.m_vf_coef;
    algo.m_optimizer->zero_grad();
    r.loss.backward();
    return r;
}

// Re-drive of one or more updates through the reference's public methods, recording everything.
template <class Algo, bool Masked>
void goldTrainScenario(const RunCfg& cfg, const std::string& tag, const std::string& outPath) {
    GoldWriter g;
    std::string home;
    {
        char cwd[4096];
        home = getcwd(cwd, sizeof cwd);
    }

    // ---- instance A: manual re-drive -------------------------------------------------------
    enterScratchWithConfig(cfg, tag + "_a");
    auto algoPtr = std::make_unique<Algo>();
    Algo& algo = *algoPtr;
    const int64_t T = algo.m_num_steps, N = algo.m_num_envs, B = algo.m_batch_size, MB = algo.m_minibatch_size;
    std::vector<torch::Tensor> params = algo.m_agent->parameters();
    {
        std::vector<int64_t> meta = { T, N, algo.m_obs_size, algo.m_action_size, algo.m_num_minibatches,
                                      algo.m_update_epochs, algo.m_max_episode_steps, algo.m_seed, cfg.updates,
                                      algo.m_anneal_lr, algo.m_use_gae, algo.m_norm_adv, algo.m_clip_vloss, Masked };
        g.addI64("meta", meta, { static_cast<int64_t>(meta.size()) });
        std::vector<float> hp = { algo.m_learning_rate, algo.m_gamma, algo.m_gae_lambda, algo.m_clip_coef,
                                  algo.m_ent_coef, algo.m_vf_coef, algo.m_max_grad_norm };
        g.addF32("hparams", hp, { static_cast<int64_t>(hp.size()) });
        std::vector<int64_t> shapes;
        for (const auto& p : params) { shapes.push_back(p.dim() > 0 ? p.size(0) : 1); shapes.push_back(p.dim() > 1 ? p.size(1) : 1); }
        g.addI64("param_shapes", shapes, { static_cast<int64_t>(params.size()), 2 });
    }
    g.add("params_init", flatParams(params));

    algo.m_threadPool->start();
    torch::Tensor next_obs, next_done = torch::zeros({ N });
    torch::Tensor next_mask;
    if constexpr (Masked) {
        next_mask = torch::ones({ N, 3 }, torch::kBool);
        next_obs = algo.initEnvs(next_mask);
    } else {
        next_obs = algo.initEnvs();
    }
    g.add("init_obs", next_obs);

    const int64_t num_updates = cfg.updates;
    for (int64_t update = 1; update <= num_updates; update++) {
        const std::string U = "u" + std::to_string(update) + "/";
        if (algo.m_anneal_lr) {
            double frac = 1.0 - (update - 1.0) / num_updates;
            double lr_now = frac * algo.m_learning_rate;
            static_cast<torch::optim::AdamWOptions&>(algo.m_optimizer->param_groups()[0].options()).lr() = lr_now;
        }
        g.addF64(U + "lr", { static_cast<torch::optim::AdamWOptions&>(algo.m_optimizer->param_groups()[0].options()).lr() }, { 1 });
        g.add(U + "params_before", flatParams(params));

        torch::Tensor reward, done;
        std::vector<torch::Tensor> entropies;
        for (int64_t step = 0; step < T; step++) {
            algo.m_obs[step] = next_obs;
            algo.m_dones[step] = next_done;
            AgentOutput out;
            if constexpr (Masked) {
                algo.m_action_masks[step] = next_mask;
                out = algo.computeActionLogic(next_obs, next_mask);
            } else {
                out = algo.computeActionLogic(next_obs);
            }
            entropies.push_back(out.entropy.detach().clone());
            algo.m_values[step] = out.value.flatten();
            if constexpr (Masked) algo.m_actions[step] = out.action;
            else algo.m_actions[step] = out.action.unsqueeze(1);
            algo.m_logprobs[step] = out.logprob;
            std::tie(next_obs, reward, done) = algo.stepEnvs(out.action.cpu());
            algo.m_rewards[step] = reward.view(-1);
            next_done = done.squeeze();
        }
        g.add(U + "obs", algo.m_obs);
        g.add(U + "actions", algo.m_actions);
        g.add(U + "logprobs", algo.m_logprobs);
        g.add(U + "rewards", algo.m_rewards);
        g.add(U + "dones", algo.m_dones);
        g.add(U + "values", algo.m_values);
        g.add(U + "rollout_entropy", torch::stack(entropies));
        if constexpr (Masked) g.add(U + "action_masks", algo.m_action_masks);
        g.add(U + "next_obs", next_obs);
        g.add(U + "next_done", next_done);  // int32 [N]
        {
            torch::NoGradGuard ng;
            g.add(U + "next_value", algo.m_agent->getValue(next_obs).reshape({ 1, -1 }));
        }
        g.addF64(U + "ep_stats", { algo.m_episode_stats->avgLength(), static_cast<double>(algo.m_episode_stats->avgReward()),
                                   static_cast<double>(algo.m_episode_stats->size()) }, { 3 });

        // Both advantage modes on the same rollout (PPO_Discrete.cpp:283-329).
        auto gae = algo.calcAdvantage(next_obs, next_done);
        g.add(U + "gae_returns", gae[0]);
        g.add(U + "gae_advantages", gae[1]);
        torch::Tensor returns = gae[0], advantages = gae[1];
        if (update == 1) {
            // The n-step branch (use_gae = false, PPO_Discrete.cpp:309-329) assigns a [1,N] tensor into the [N] row
            // returns[T-1] (next_return = next_value, :318,:324), which LibTorch rejects: record what the reference does.
            algo.m_use_gae = false;
            int64_t threw = 0;
            try {
                auto nstep = algo.calcAdvantage(next_obs, next_done);
                g.add(U + "nstep_returns", nstep[0]);
                g.add(U + "nstep_advantages", nstep[1]);
            } catch (const std::exception& ex) {
                threw = 1;
                std::cerr << "[ref_harness] " << tag << ": reference n-step branch throws: "
                          << std::string(ex.what()).substr(0, 90) << "\n";
            }
            algo.m_use_gae = true;
            g.addI64(U + "nstep_branch_throws", { threw }, { 1 });
        }

        torch::Tensor b_obs = algo.m_obs.reshape({ B, algo.m_obs_size });
        torch::Tensor b_logprobs = algo.m_logprobs.reshape(-1);
        torch::Tensor b_actions = Masked ? algo.m_actions.reshape({ B, algo.m_action_size }) : algo.m_actions.reshape(-1);
        torch::Tensor b_advantages = advantages.reshape(-1);
        torch::Tensor b_returns = returns.reshape(-1);
        torch::Tensor b_values = algo.m_values.reshape(-1);
        torch::Tensor b_masks;
        if constexpr (Masked) b_masks = algo.m_action_masks.reshape({ -1, algo.m_action_masks.sizes().back() });
        std::vector<float>().swap(algo.m_clipfracs);

        std::vector<torch::Tensor> perms;
        std::vector<double> scal;  // per optimizer step: pg, v, ent, kl, clipfrac, loss, total_norm
        int64_t k = 0;
        const int64_t stepsPerUpdate = algo.m_update_epochs * ((B + MB - 1) / MB);
        torch::Tensor pg_loss, v_loss, entropy_loss, loss, approx_kl;
        for (int64_t epoch = 0; epoch < algo.m_update_epochs; epoch++) {
            torch::Tensor b_inds = torch::randperm(B);
            perms.push_back(b_inds.clone());
            for (int64_t start = 0; start < B; start += MB) {
                torch::Tensor mb = b_inds.index({ torch::indexing::Slice(start, start + MB) });
                MinibatchOut r = minibatchLossBackward<Algo, Masked>(algo, mb, b_obs, b_masks, b_actions, b_logprobs, b_advantages, b_returns, b_values);
                AgentOutput& o = r.o;
                pg_loss = r.pg_loss; v_loss = r.v_loss; entropy_loss = r.entropy_loss; loss = r.loss; approx_kl = r.approx_kl;
                torch::Tensor nv = o.value.view(-1);
                const bool dump = (update == 1) && (k == 0 || k == 1 || k == stepsPerUpdate - 1);
                const std::string K = U + "k" + std::to_string(k) + "/";
                if (dump) {
                    g.add(K + "grads", flatParams(params, true));
                    g.add(K + "newlogprob", o.logprob);
                    g.add(K + "newvalue", nv);
                    g.add(K + "entropy", o.entropy);
                }
                double total_norm = torch::nn::utils::clip_grad_norm_(params, algo.m_max_grad_norm);
                algo.m_optimizer->step();
                if (dump) {
                    g.add(K + "params_after", flatParams(params));
                    auto mv = flatAdamState(*algo.m_optimizer, params);
                    g.add(K + "exp_avg", mv[0]);
                    g.add(K + "exp_avg_sq", mv[1]);
                }
                scal.insert(scal.end(), { pg_loss.item<double>(), v_loss.item<double>(), entropy_loss.item<double>(),
                                          approx_kl.item<double>(), static_cast<double>(algo.m_clipfracs.back()),
                                          loss.item<double>(), total_norm });
                k++;
            }
        }
        g.add(U + "perms", torch::stack(perms));
        g.addF64(U + "step_scalars", scal, { k, 7 });
        torch::Tensor var_y = b_returns.var();
        torch::Tensor ev = 1 - ((b_returns - b_values).var() / var_y);
        g.addF64(U + "explained_var", { ev.item<double>() }, { 1 });
        g.add(U + "params_after", flatParams(params));
    }
    algo.m_threadPool->stop();
    torch::Tensor finalManual = flatParams(params);

    // ---- instance B: the reference's own train() on the same config -------------------------
    int64_t certified = -1;  // -1 = not applicable (MountainCar resets are unseedable, MountainCar.cpp:79-88)
    if constexpr (!Masked) {
        enterScratchWithConfig(cfg, tag + "_b");
        Algo twin;
        twin.train();
        torch::Tensor finalReal = flatParams(twin.m_agent->parameters());
        certified = torch::equal(finalReal, finalManual) ? 1 : 0;
        std::cerr << "[ref_harness] " << tag << ": manual re-drive vs reference train(): "
                  << (certified ? "BIT-IDENTICAL" : "MISMATCH") << " (max abs diff "
                  << (finalReal - finalManual).abs().max().item<double>() << ")\n";
        g.add("params_after_reference_train", finalReal);
    }
    g.addI64("certified_bitwise", { certified }, { 1 });
    if (chdir(home.c_str()) != 0) throw std::runtime_error("chdir back failed");
    g.save(outPath);
    if (certified == 0) throw std::runtime_error("re-drive of train() is not bit-identical to the reference's train()");
}

void goldResetStream(const std::string& outPath) {
    GoldWriter g;
    for (int64_t seed : { 1, 2, 3, 12345 }) {
        CartPole env(seed);
        std::vector<float> v;
        for (int k = 0; k < 64; k++) {
            std::vector<float> s = env.reset();
            v.insert(v.end(), s.begin(), s.end());
        }
        g.addF32("seed" + std::to_string(seed), v, { 64, 4 });
    }
    g.save(outPath);
}

float nudge(float v, int ulps) {
    for (int i = 0; i < std::abs(ulps); i++) v = std::nextafter(v, ulps > 0 ? INFINITY : -INFINITY);
    return v;
}

void goldCartPoleTransitions(const std::string& outPath) {
    GoldWriter g;
    const int64_t n = 8192;
    std::mt19937_64 rng(20240611);
    auto U = [&](float a, float b) { return std::uniform_real_distribution<float>(a, b)(rng); };
    std::vector<float> s0(n * 4), s1(n * 4), rew(n);
    std::vector<int64_t> act(n), term(n);
    const float thr = static_cast<float>(12 * 2 * M_PI / 360);
    for (int64_t i = 0; i < n; i++) {
        float x, xd, th, thd;
        int kind = static_cast<int>(i % 8);
        if (kind < 4) { x = U(-2.6f, 2.6f); xd = U(-3.f, 3.f); th = U(-0.25f, 0.25f); thd = U(-3.5f, 3.5f); }
        else if (kind < 6) { x = U(-0.05f, 0.05f); xd = U(-0.05f, 0.05f); th = U(-0.05f, 0.05f); thd = U(-0.05f, 0.05f); }
        else if (kind == 6) {  // threshold edges: zero velocity keeps the position exactly
            int u = static_cast<int>((i / 8) % 7) - 3;
            bool onX = ((i / 56) % 2) == 0;
            float sgn = ((i / 112) % 2) ? -1.f : 1.f;
            x = onX ? sgn * nudge(2.4f, u) : U(-1.f, 1.f);
            th = onX ? U(-0.1f, 0.1f) : sgn * nudge(thr, u);
            xd = onX ? 0.f : U(-1.f, 1.f);
            thd = onX ? U(-1.f, 1.f) : 0.f;
        } else { x = U(-2.f, 2.f); xd = U(-2.f, 2.f); th = U(-3.0f, 3.0f); thd = U(-8.f, 8.f); }
        int64_t a = static_cast<int64_t>(rng() % 3);  // 2 exercises "any non-zero action pushes right" (CartPole.cpp:54-57)
        CartPole env(1);
        env.state = { x, xd, th, thd };
        env.terminated = false;
        auto [ns, r, t, info] = env.step(a);
        (void)info;
        for (int j = 0; j < 4; j++) { s0[i * 4 + j] = (j == 0 ? x : j == 1 ? xd : j == 2 ? th : thd); s1[i * 4 + j] = ns[j]; }
        rew[i] = r; act[i] = a; term[i] = t ? 1 : 0;
    }
    g.addF32("state", s0, { n, 4 });
    g.addI64("action", act, { n });
    g.addF32("next_state", s1, { n, 4 });
    g.addF32("reward", rew, { n });
    g.addI64("terminated", term, { n });
    g.save(outPath);
}

void goldMountainCarTransitions(const std::string& outPath) {
    GoldWriter g;
    const int64_t n = 8192;
    std::mt19937_64 rng(777);
    auto U = [&](float a, float b) { return std::uniform_real_distribution<float>(a, b)(rng); };
    std::vector<float> s0(n * 2), s1(n * 2), rew(n);
    std::vector<int64_t> act(n), term(n);
    for (int64_t i = 0; i < n; i++) {
        float p = U(-1.2f, 0.6f), v = U(-0.07f, 0.07f);
        int kind = static_cast<int>(i % 8);
        if (kind == 5) { p = -1.2f; v = U(-0.07f, 0.0f); }               // left wall (MountainCar.cpp:40-42)
        if (kind == 6) { p = U(0.45f, 0.6f); v = U(-0.01f, 0.07f); }     // goal edge (:44)
        if (kind == 7) { p = U(-0.6f, -0.4f); v = 0.f; }                 // reset states
        int64_t a = static_cast<int64_t>(rng() % 3);
        MountainCar env;
        env.state = { p, v };
        auto [ns, r, t, info] = env.step(a);
        (void)info;
        s0[i * 2] = p; s0[i * 2 + 1] = v; s1[i * 2] = ns[0]; s1[i * 2 + 1] = ns[1];
        rew[i] = r; act[i] = a; term[i] = t ? 1 : 0;
    }
    g.addF32("state", s0, { n, 2 });
    g.addI64("action", act, { n });
    g.addF32("next_state", s1, { n, 2 });
    g.addF32("reward", rew, { n });
    g.addI64("terminated", term, { n });
    g.save(outPath);
}

void goldDistributions(const std::string& outPath) {
    GoldWriter g;
    torch::manual_seed(99);
    auto dev = std::make_shared<torch::Device>(torch::kCPU);
    for (int64_t A : { 1, 2, 3, 6 }) {
        const std::string P = "cat" + std::to_string(A) + "/";
        torch::Tensor logits = torch::randn({ 96, A }) * 2.0;
        if (A >= 2) logits.index_put_({ 0 }, torch::full({ A }, 30.0f).index_put_({ 0 }, -30.0f));  // extreme row
        Categorical c(logits, dev);
        torch::Tensor value = torch::randint(A, { 96 });
        g.add(P + "logits", logits);
        g.add(P + "m_logits", c.m_logits);
        g.add(P + "m_probs", c.m_probs);
        g.add(P + "value", value);
        g.add(P + "log_prob", c.log_prob(value));
        g.add(P + "entropy", c.entropy());
        g.add(P + "mode", c.mode());
    }
    for (int64_t A : { 2, 3, 6 }) {
        const std::string P = "masked" + std::to_string(A) + "/";
        torch::Tensor logits = torch::randn({ 96, A }) * 2.0;
        torch::Tensor mask = torch::rand({ 96, A }) > 0.4;
        torch::Tensor keep = torch::randint(A, { 96 });
        mask.scatter_(1, keep.unsqueeze(1), torch::ones({ 96, 1 }, torch::kBool));  // >= 1 valid action
        CategoricalMasked c(logits, mask, dev);
        g.add(P + "logits", logits);
        g.add(P + "mask", mask);
        g.add(P + "m_logits", c.m_logits);
        g.add(P + "m_probs", c.m_probs);
        g.add(P + "value", keep);
        g.add(P + "log_prob", c.log_prob(keep));
        g.add(P + "entropy", c.entropy());
        g.add(P + "mode", c.mode());
    }
    g.save(outPath);
}

// Multi-head masked agent: the split by m_actionSpace (Agent.cpp:140-141) with more than one head.
void goldMultiHeadAgent(const std::string& outPath) {
    GoldWriter g;
    torch::manual_seed(5);
    auto dev = std::make_shared<torch::Device>(torch::kCPU);
    const int64_t O = 4, n = 64;  // CartPole-sized observations so the context-level C-ABI can replay it
    std::vector<int64_t> heads = { 3, 3, 3, 2 };
    Agent agent(O, 11, dev);
    agent.m_actionSpace = heads;
    torch::Tensor x = torch::randn({ n, O });
    torch::Tensor mask = torch::rand({ n, 11 }) > 0.35;
    std::vector<torch::Tensor> acts;
    int64_t off = 0;
    for (int64_t h : heads) {
        torch::Tensor a = torch::randint(h, { n });
        mask.scatter_(1, (a + off).unsqueeze(1), torch::ones({ n, 1 }, torch::kBool));
        acts.push_back(a);
        off += h;
    }
    torch::Tensor action = torch::stack(acts);  // [H, n] as getActionAndValueMasked indexes action[i] (Agent.cpp:160-163)
    AgentOutput o;
    {
        torch::NoGradGuard ng;
        o = agent.getActionAndValueMasked(x, mask, action);
    }
    g.addI64("heads", heads, { 4 });
    g.add("params", flatParams(agent.parameters()));
    std::vector<int64_t> shapes;
    for (const auto& p : agent.parameters()) { shapes.push_back(p.dim() > 0 ? p.size(0) : 1); shapes.push_back(p.dim() > 1 ? p.size(1) : 1); }
    g.addI64("param_shapes", shapes, { static_cast<int64_t>(shapes.size() / 2), 2 });
    g.add("x", x);
    g.add("mask", mask);
    g.add("action_hn", action);
    g.add("action_out", o.action);
    g.add("logprob", o.logprob);
    g.add("entropy", o.entropy);
    g.add("value", o.value);
    g.save(outPath);
}

int benchReference(int64_t numEnvs, int64_t numSteps, int64_t updates, int64_t threads) {
    RunCfg c;
    c.num_envs = numEnvs; c.num_steps = numSteps; c.updates = updates;
    enterScratchWithConfig(c, "bench");
    PPO_Discrete algo;
    unsigned poolThreads = std::thread::hardware_concurrency();
    if (threads > 0) {
        // the pool is a public member built in the constructor with hardware_concurrency() threads (PPO_Discrete.cpp:40) and started by
        // train() (:488): a narrower pool of the reference's own class, and LibTorch's intra-op threads capped to match
        algo.m_threadPool = std::make_shared<ThreadPool>(threads);
        at::set_num_threads(static_cast<int>(threads));
        poolThreads = static_cast<unsigned>(threads);
    }
    auto t0 = std::chrono::steady_clock::now();
    algo.train();
    auto t1 = std::chrono::steady_clock::now();
    double sec = std::chrono::duration<double>(t1 - t0).count();
    double steps = static_cast<double>(updates) * numEnvs * numSteps;
    std::printf("REF_BENCH {\"num_envs\": %ld, \"num_steps\": %ld, \"updates\": %ld, \"seconds\": %.6f, "
                "\"env_steps_per_sec\": %.3f, \"threads\": %u, \"hardware_concurrency\": %u, \"torch_threads\": %d}\n",
                (long)numEnvs, (long)numSteps, (long)updates, sec, steps / sec, poolThreads, std::thread::hardware_concurrency(), at::get_num_threads());
    return 0;
}

// Host-side fixtures: what the reference PRINTS and PARSES, for crafted inputs that
// ppo-libtorch_amd/host/tests/host_facade_test.cpp feeds to the facade's own printPPOResults / PPOUtils.
//   console_table.txt     the SB3-style table of PPO_Discrete::printPPOResults (PPO_Discrete.cpp:700-774) for four calls made in this order in ONE
//                         process (stream manipulators persist between calls, as they do in train()): first update with episode statistics,
//                         a later update with them, a later update without, a first update without
//   host_utils.txt        PPOUtils::getLoadFromSteps / isNumber / getVectorMean (Utils.cpp:5-60) on a list of inputs
void goldHost(const std::string& outDir) {
    RunCfg c;   // 8 envs x 32 steps: m_batch_size = 256, update_epochs = 10, clip_coef = 0.2
    enterScratchWithConfig(c, "hostgold");
    PPO_Discrete algo;
    static_cast<torch::optim::AdamWOptions&>(algo.m_optimizer->param_groups()[0].options()).lr(0.00075);
    algo.m_clipfracs = { 0.125f, 0.0625f, 0.25f };
    torch::Tensor kl = torch::tensor(0.00123456789f), ent = torch::tensor(-1.17549435e-38f), ev = torch::tensor(0.1762397289f);
    torch::Tensor loss = torch::tensor(24.916658401f), pg = torch::tensor(-0.007237161f), vl = torch::tensor(27.17522430f);
    std::stringstream ss;
    std::streambuf* old = std::cout.rdbuf(ss.rdbuf());
    algo.m_episode_stats = std::make_unique<CircularBuffer>(100);
    algo.m_episode_stats->add(21.0f, 22); algo.m_episode_stats->add(13.0f, 14); algo.m_episode_stats->add(-1.0f, 1); algo.m_episode_stats->add(499.0f, 500);
    algo.printPPOResults(1, 256, std::chrono::milliseconds(123), std::chrono::milliseconds(4567), kl, ent, ev, loss, pg, vl);
    algo.printPPOResults(2, 512, std::chrono::milliseconds(97), std::chrono::milliseconds(12345), kl, ent, ev, loss, pg, vl);
    algo.m_episode_stats = std::make_unique<CircularBuffer>(100);
    algo.printPPOResults(3, 768, std::chrono::milliseconds(97), std::chrono::milliseconds(23456), kl, ent, ev, loss, pg, vl);
    algo.printPPOResults(1, 256, std::chrono::milliseconds(123), std::chrono::milliseconds(999), kl, ent, ev, loss, pg, vl);
    std::cout.rdbuf(old);
    { std::ofstream f(outDir + "/console_table.txt", std::ios::binary); f << ss.str(); }
    std::ofstream u(outDir + "/host_utils.txt", std::ios::binary);
    const char* names[] = { "./ModelCheckpoints/PPO_Agent_99840_steps.pt", "PPO_Agent_0_steps.pt", "./OptimizerCheckpoints/PPO_Optimizer_123456789_steps.pt",
                            "PPO_Agent_steps.pt", "PPO_Agent_12ab_steps.pt", "/a/b/PPO_Agent_77", "./Models/PPO_Agent_5000000_steps.pt" };
    for (const char* n : names) {
        const std::string head = std::string(n).find("Optimizer") != std::string::npos ? "PPO_Optimizer_" : "PPO_Agent_";
        const std::string r = PPOUtils::getLoadFromSteps(n, head);
        u << "steps|" << n << "|" << head << "|" << r << "|" << (PPOUtils::isNumber(r) ? 1 : 0) << "\n";
    }
    const char* nums[] = { "123", "", "12a", "007", " 1", "-5" };
    for (const char* n : nums) u << "isnum|" << n << "|" << (PPOUtils::isNumber(n) ? 1 : 0) << "\n";
    char buf[64];
    std::vector<std::vector<float>> vecs = { { 0.125f, 0.0625f, 0.25f }, { 1.0f }, { 0.1f, 0.2f, 0.3f, 0.4f } };
    for (const auto& v : vecs) {
        const float m = PPOUtils::getVectorMean(v);
        uint32_t bits; std::memcpy(&bits, &m, 4);
        std::snprintf(buf, sizeof buf, "%08x", bits);
        u << "mean|";
        for (size_t i = 0; i < v.size(); i++) { uint32_t b; std::memcpy(&b, &v[i], 4); char t[16]; std::snprintf(t, sizeof t, "%08x", b); u << (i ? "," : "") << t; }
        u << "|" << buf << "\n";
    }
}

// (8) What the reference's constructor prints (getArgs' "Using config file ..." lines, the fallback message, the checkpoint message) and the
// hyper-parameters it ends up with, for a full PPOConfig.toml, a partial one and none at all (PPO_Discrete.cpp:10-105, 107-255).  The
// fixture carries the TOML text itself, so the facade's test feeds its own constructor exactly the same file.
void goldGetArgs(const std::string& outDir) {
    const std::string full =
        "[environment]\nobs_size = 4\naction_size = 2\nmax_episode_steps = 321\n\n"
        "[general]\nseed = 7\ntotal_timesteps = 4096\nuse_cuda = false\ntorch_deterministic = true\ncheckpoint_updates = 9\n\n"
        "[ppo]\nlearning_rate = 0.00075\nnum_envs = 4\nnum_steps = 16\nanneal_lr = true\nuse_gae = false\ngamma = 0.97\ngae_lambda = 0.9\n"
        "num_minibatches = 2\nupdate_epochs = 3\nnorm_adv = false\nclip_coef = 0.15\nclip_vloss = false\nent_coef = 0.001\nvf_coef = 0.25\nmax_grad_norm = 1.5\n";
    const std::string partial =
        "# only some keys: everything else keeps the constructor's default\n[general]\nseed = 11\n\n[ppo]\nnum_envs = 2\nnum_steps = 8\nnum_minibatches = 4\ngamma = 0.5\nanneal_lr = true\n";
    // PPO_MultiDiscrete reads two more keys, between action_size and max_episode_steps (PPO_MultiDiscrete.cpp:136-144); PPO_Discrete ignores them
    const std::string multi =
        "[environment]\nobs_size = 2\naction_size = 3\naction_high = 2.5\naction_low = -0.5\nmax_episode_steps = 150\n\n"
        "[general]\nseed = 3\ntotal_timesteps = 2048\n\n[ppo]\nnum_envs = 4\nnum_steps = 16\nnum_minibatches = 2\nent_coef = 0.02\n";
    struct V { const char* name; const std::string* toml; bool multi; };
    const V variants[] = { { "full", &full, false }, { "partial", &partial, false }, { "none", nullptr, false }, { "multidiscrete", &multi, true },
                           { "discrete_ignores_action_bounds", &multi, false } };
    std::ofstream out(outDir + "/getargs.txt", std::ios::binary);
    for (const V& v : variants) {
        std::string dir = makeScratchDir(std::string("getargs_") + v.name);
        if (chdir(dir.c_str()) != 0) throw std::runtime_error("chdir failed");
        if (v.toml) { std::ofstream f("PPOConfig.toml", std::ios::binary); f << *v.toml; }
        std::stringstream ss;
        std::cout.copyfmt(std::ios(nullptr));   // the constructor is the first thing the reference's driver runs: pristine stream state
        std::streambuf* old = std::cout.rdbuf(ss.rdbuf());
        std::unique_ptr<PPO_Discrete> algo;
        std::unique_ptr<PPO_MultiDiscrete> malgo;
        try { if (v.multi) malgo = std::make_unique<PPO_MultiDiscrete>(); else algo = std::make_unique<PPO_Discrete>(); } catch (...) { std::cout.rdbuf(old); throw; }
        std::cout.rdbuf(old);
        if (v.multi) {   // same field list, read off the other class
            auto& a = *malgo;
            out << "== variant " << v.name << "\n-- toml\n" << *v.toml << "-- stdout\n" << ss.str() << "-- fields\n" << std::setprecision(9);
            out << "m_obs_size=" << a.m_obs_size << "\nm_action_size=" << a.m_action_size << "\nm_max_episode_steps=" << a.m_max_episode_steps
                << "\nm_seed=" << a.m_seed << "\nm_total_timesteps=" << a.m_total_timesteps << "\nm_use_cuda=" << (a.m_use_cuda ? 1 : 0)
                << "\nm_torch_deterministic=" << (a.m_torch_deterministic ? 1 : 0) << "\nm_checkpoint_updates=" << a.m_checkpoint_updates
                << "\nm_learning_rate=" << a.m_learning_rate << "\nm_num_envs=" << a.m_num_envs << "\nm_num_steps=" << a.m_num_steps
                << "\nm_anneal_lr=" << (a.m_anneal_lr ? 1 : 0) << "\nm_use_gae=" << (a.m_use_gae ? 1 : 0) << "\nm_gamma=" << a.m_gamma
                << "\nm_gae_lambda=" << a.m_gae_lambda << "\nm_num_minibatches=" << a.m_num_minibatches << "\nm_update_epochs=" << a.m_update_epochs
                << "\nm_norm_adv=" << (a.m_norm_adv ? 1 : 0) << "\nm_clip_coef=" << a.m_clip_coef << "\nm_clip_vloss=" << (a.m_clip_vloss ? 1 : 0)
                << "\nm_ent_coef=" << a.m_ent_coef << "\nm_vf_coef=" << a.m_vf_coef << "\nm_max_grad_norm=" << a.m_max_grad_norm
                << "\nm_batch_size=" << a.m_batch_size << "\nm_minibatch_size=" << a.m_minibatch_size << "\n";
            continue;
        }
        out << "== variant " << v.name << "\n-- toml\n" << (v.toml ? *v.toml : std::string("(none)\n")) << "-- stdout\n" << ss.str() << "-- fields\n";
        out << std::setprecision(9);
        out << "m_obs_size=" << algo->m_obs_size << "\nm_action_size=" << algo->m_action_size << "\nm_max_episode_steps=" << algo->m_max_episode_steps
            << "\nm_seed=" << algo->m_seed << "\nm_total_timesteps=" << algo->m_total_timesteps << "\nm_use_cuda=" << (algo->m_use_cuda ? 1 : 0)
            << "\nm_torch_deterministic=" << (algo->m_torch_deterministic ? 1 : 0) << "\nm_checkpoint_updates=" << algo->m_checkpoint_updates
            << "\nm_learning_rate=" << algo->m_learning_rate << "\nm_num_envs=" << algo->m_num_envs << "\nm_num_steps=" << algo->m_num_steps
            << "\nm_anneal_lr=" << (algo->m_anneal_lr ? 1 : 0) << "\nm_use_gae=" << (algo->m_use_gae ? 1 : 0) << "\nm_gamma=" << algo->m_gamma
            << "\nm_gae_lambda=" << algo->m_gae_lambda << "\nm_num_minibatches=" << algo->m_num_minibatches << "\nm_update_epochs=" << algo->m_update_epochs
            << "\nm_norm_adv=" << (algo->m_norm_adv ? 1 : 0) << "\nm_clip_coef=" << algo->m_clip_coef << "\nm_clip_vloss=" << (algo->m_clip_vloss ? 1 : 0)
            << "\nm_ent_coef=" << algo->m_ent_coef << "\nm_vf_coef=" << algo->m_vf_coef << "\nm_max_grad_norm=" << algo->m_max_grad_norm
            << "\nm_batch_size=" << algo->m_batch_size << "\nm_minibatch_size=" << algo->m_minibatch_size << "\n";
    }
}

// (9) A short train() run of the reference with checkpoints, then a second instance resuming in the same directory (PPO_Discrete.cpp:485-690,
// 782-835): every console line that is not part of the per-update table (the table carries wall-clock numbers), the files the run leaves
// behind, and the step the second instance resumes from.
static std::string nonTableLines(const std::string& all) {
    std::istringstream is(all);
    std::string l, out;
    while (std::getline(is, l)) {
        if (l.empty() || l[0] == '-') continue;
        if (l[0] == '|') {   // table rows whose value does not depend on the wall clock or on the random stream
            bool keep = false;
            for (const char* k : { "iterations", "total_timesteps", "clip_range", "learning_rate", "n_updates" }) keep |= l.find(std::string("|    ") + k + " ") == 0;
            if (!keep) continue;
        }
        out += l + "\n";
    }
    return out;
}
static std::string listFiles() {
    std::vector<std::string> names;
    for (const char* d : { "ModelCheckpoints", "OptimizerCheckpoints", "Models" })
        if (std::filesystem::exists(d)) for (auto const& e : std::filesystem::directory_iterator(d)) names.push_back(std::string(d) + "/" + e.path().filename().string());
    std::sort(names.begin(), names.end());
    std::string out;
    for (auto& n : names) out += n + "\n";
    return out;
}
template <class Algo>
void goldTrainRun(const std::string& outFile, const std::string& toml) {
    std::string dir = makeScratchDir("trainrun");
    if (chdir(dir.c_str()) != 0) throw std::runtime_error("chdir failed");
    { std::ofstream f("PPOConfig.toml", std::ios::binary); f << toml; }
    std::ofstream out(outFile, std::ios::binary);
    out << "-- toml\n" << toml;
    auto captured = [&](auto&& fn) {
        std::stringstream ss;
        std::cout.copyfmt(std::ios(nullptr));
        std::streambuf* old = std::cout.rdbuf(ss.rdbuf());
        try { fn(); } catch (...) { std::cout.rdbuf(old); throw; }
        std::cout.rdbuf(old);
        return ss.str();
    };
    {
        std::unique_ptr<Algo> algo;
        const std::string c1 = captured([&] { algo = std::make_unique<Algo>(); });
        const std::string t1 = captured([&] { algo->train(); });
        out << "-- phase1 constructor\n" << nonTableLines(c1) << "-- phase1 train\n" << nonTableLines(t1) << "-- phase1 files\n" << listFiles();
    }
    {   // a second instance in the same directory: the reference's resume.  Built against libstdc++ it THROWS: loadPolicyFromCheckpoint compares
        // every file's last_write_time with a default-constructed file_time_type (:801-806), and libstdc++'s file_clock epoch is the year 2174,
        // so no file is ever "newer", the name stays empty and PPOUtils::getLoadFromSteps("") runs off the string.  (MSVC's epoch is 1601:
        // there it resumes.)  Recorded as a fact about the reference on this platform; the facade's resume is tested on its own.
        std::string what = "(no exception)";
        std::string c2;
        try {
            std::unique_ptr<Algo> algo;
            c2 = captured([&] { algo = std::make_unique<Algo>(); });
        } catch (const std::exception& ex) { what = ex.what(); }
        out << "-- phase2 constructor\n" << nonTableLines(c2) << "-- phase2 resume exception\n" << what << "\n";
    }
}

// The checkpoint files themselves (PPO_Discrete.cpp:662-685, 779-788): a short train() run, then the two `.pt` archives it left under
// ./Models/ are copied out as fixtures together with what they hold (parameters in Agent::parameters() order, AdamW step counts and
// moments), so the host facade's reader of the reference's files can be checked value for value.
template <class Algo>
void goldCheckpointFiles(const std::string& outDir, const std::string& tag, const std::string& toml, int64_t total_timesteps) {
    std::string dir = makeScratchDir("ptgold_" + tag);
    if (chdir(dir.c_str()) != 0) throw std::runtime_error("chdir failed");
    { std::ofstream f("PPOConfig.toml", std::ios::binary); f << toml; }
    std::stringstream sink;
    std::streambuf* old = std::cout.rdbuf(sink.rdbuf());
    std::unique_ptr<Algo> algo;
    try { algo = std::make_unique<Algo>(); algo->train(); } catch (...) { std::cout.rdbuf(old); throw; }
    std::cout.rdbuf(old);
    const std::string stem = "./Models/PPO_";
    const std::string sfx = "_" + std::to_string(total_timesteps) + "_steps.pt";
    std::filesystem::copy_file(stem + "Agent" + sfx, outDir + "/ref_" + tag + "_agent.pt", std::filesystem::copy_options::overwrite_existing);
    std::filesystem::copy_file(stem + "Optimizer" + sfx, outDir + "/ref_" + tag + "_optimizer.pt", std::filesystem::copy_options::overwrite_existing);
    GoldWriter g;
    std::vector<torch::Tensor> params = algo->m_agent->parameters();
    g.add("params", flatParams(params));
    auto mv = flatAdamState(*algo->m_optimizer, params);
    g.add("exp_avg", mv[0]);
    g.add("exp_avg_sq", mv[1]);
    std::vector<int64_t> steps, numels;
    for (const auto& p : params) {
        auto& st = static_cast<torch::optim::AdamWParamState&>(*algo->m_optimizer->state().at(p.unsafeGetTensorImpl()));
        steps.push_back(st.step());
        numels.push_back(p.numel());
    }
    g.addI64("steps", steps, { (int64_t)steps.size() });
    g.addI64("numels", numels, { (int64_t)numels.size() });
    g.addF64("lr", { static_cast<torch::optim::AdamWOptions&>(algo->m_optimizer->param_groups()[0].options()).lr() }, { 1 });
    g.save(outDir + "/ref_" + tag + "_checkpoint_values.pgld");
}

// The reference's own load calls on given files: an Agent and an AdamW built as PPO_Discrete's constructor builds them (:72-78), then
// torch::load(m_agent, file) (:812) and torch::load(*m_optimizer, file) (:834).  What they now hold is written out, so archives produced
// by the host facade's writer can be shown to load into the reference value for value.  (loadPolicyFromCheckpoint itself cannot be
// driven here: built against libstdc++ it throws before it reaches these calls, see goldTrainRun.)
void loadCheckpointFiles(const std::string& agentFile, const std::string& optimizerFile, int64_t obs, int64_t act, const std::string& outFile) {
    auto device = std::make_shared<torch::Device>(torch::kCPU);
    auto agent = std::make_shared<Agent>(obs, act, device);
    auto optimizer = std::make_shared<torch::optim::AdamW>(agent->parameters(), torch::optim::AdamWOptions(0.5).eps(1e-5f));
    torch::load(agent, agentFile);
    torch::load(*optimizer, optimizerFile);
    GoldWriter g;
    std::vector<torch::Tensor> params = agent->parameters();
    g.add("params", flatParams(params));
    auto mv = flatAdamState(*optimizer, params);
    g.add("exp_avg", mv[0]);
    g.add("exp_avg_sq", mv[1]);
    std::vector<int64_t> steps;
    for (const auto& p : params) steps.push_back(static_cast<torch::optim::AdamWParamState&>(*optimizer->state().at(p.unsafeGetTensorImpl())).step());
    g.addI64("steps", steps, { (int64_t)steps.size() });
    auto& opt = static_cast<torch::optim::AdamWOptions&>(optimizer->param_groups()[0].options());
    g.addF64("lr", { opt.lr() }, { 1 });
    g.addF64("eps", { opt.eps() }, { 1 });
    g.addF64("weight_decay", { opt.weight_decay() }, { 1 });
    g.save(outFile);
}

// (11) Headline-size pin: BASELINE.json configs[1] (CartPole, 4096 envs x 128 steps) and configs[3] (MountainCar, 8192 x 128, masked) driven through
// the reference's own components in train()'s order (PPO_Discrete.cpp:524-548, 274-306, 554-648) with everything random INJECTED from a counter hash
// both sides can regenerate (tests/test_gpu_headline_ref.py): the actions of the rollout, the permutations of the update, MountainCar's initial
// positions.  Tensors of this size cannot be committed, so the fixture carries
//   * CRC-32s of what must match bit for bit: obs / rewards / dones / next_obs / next_done of the rollout, and advantages / returns of calcAdvantage
//     on SYNTHETIC values (hash-made floats in m_values; the critic's head zeroed with bias 0.25, so next_value is exactly 0.25 on both sides);
//   * what matches within fp32 noise: binary64 sums and a strided sample of logprobs / values / the real advantages, the 40 x 7 per-step scalars of the
//     update, parameters before and after.
namespace hl {
inline uint64_t mix64(uint64_t x) {   // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
inline float unit24(uint64_t h) { return static_cast<float>(static_cast<uint32_t>(h >> 40)) * 5.9604644775390625e-8f; }   // [0, 1), 24 bits: exact
uint32_t crc32(const void* data, size_t n) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; }
        init = true;
    }
    uint32_t c = 0xFFFFFFFFu;
    const unsigned char* p = static_cast<const unsigned char*>(data);
    for (size_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}
uint32_t crcOf(const torch::Tensor& t) { torch::Tensor c = t.detach().cpu().contiguous(); return crc32(c.data_ptr(), static_cast<size_t>(c.nbytes())); }
constexpr uint64_t SEED_ACT = 0x1111ull << 32, SEED_PERM = 0x2222ull << 32, SEED_VAL = 0x3333ull << 32, SEED_POS = 0x4444ull << 32;
}  // namespace hl

template <class Algo, bool Masked>
void goldHeadline(RunCfg cfg, const std::string& tag, const std::string& outPath) {
    GoldWriter g;
    cfg.updates = 1;
    enterScratchWithConfig(cfg, "headline_" + tag);
    auto algoPtr = std::make_unique<Algo>();
    Algo& algo = *algoPtr;
    const int64_t T = algo.m_num_steps, N = algo.m_num_envs, B = algo.m_batch_size, MB = algo.m_minibatch_size, A = algo.m_action_size;
    std::vector<torch::Tensor> params = algo.m_agent->parameters();
    {
        std::vector<int64_t> meta = { T, N, algo.m_obs_size, A, algo.m_num_minibatches, algo.m_update_epochs, algo.m_max_episode_steps, algo.m_seed, 1,
                                      algo.m_anneal_lr, algo.m_use_gae, algo.m_norm_adv, algo.m_clip_vloss, Masked };
        g.addI64("meta", meta, { static_cast<int64_t>(meta.size()) });
        std::vector<float> hp = { algo.m_learning_rate, algo.m_gamma, algo.m_gae_lambda, algo.m_clip_coef, algo.m_ent_coef, algo.m_vf_coef, algo.m_max_grad_norm };
        g.addF32("hparams", hp, { static_cast<int64_t>(hp.size()) });
    }
    g.add("params_before", flatParams(params));
    algo.m_threadPool->start();
    torch::Tensor next_obs, next_done = torch::zeros({ N });
    torch::Tensor next_mask;
    if constexpr (Masked) {
        next_mask = torch::ones({ N, 3 }, torch::kBool);
        next_obs = algo.initEnvs(next_mask);
        // MountainCar::reset draws from std::random_device (MountainCar.cpp:79-88): the initial positions are injected (hash-made, the reference's range)
        next_obs = next_obs.clone();
        for (int64_t n = 0; n < N; n++) {
            const float a = hl::unit24(hl::mix64(hl::SEED_POS + static_cast<uint64_t>(n)));
            const float b = 0.2f * a;
            const float p0 = -0.6f + b;
            algo.m_envs[n]->state = { p0, 0.0f };
            algo.m_envs[n]->episode_length = 0;
            algo.m_envs[n]->episode_reward = 0.0f;
            next_obs[n][0] = p0; next_obs[n][1] = 0.0f;
        }
    } else {
        next_obs = algo.initEnvs();
    }
    g.addI64("crc_init_obs", { hl::crcOf(next_obs) }, { 1 });

    // ---- rollout with injected actions (PPO_Discrete.cpp:524-548) ----
    torch::Tensor reward, done;
    {
        torch::NoGradGuard ng;
        for (int64_t step = 0; step < T; step++) {
            algo.m_obs[step] = next_obs;
            algo.m_dones[step] = next_done;
            std::vector<int64_t> av(N);
            for (int64_t n = 0; n < N; n++) av[n] = static_cast<int64_t>(hl::mix64(hl::SEED_ACT + static_cast<uint64_t>(step * N + n)) % static_cast<uint64_t>(A));
            torch::Tensor action = torch::from_blob(av.data(), { N }, torch::kInt64).clone();
            AgentOutput out;
            if constexpr (Masked) {
                algo.m_action_masks[step] = next_mask;
                out = algo.m_agent->getActionAndValueMasked(next_obs, next_mask, action.unsqueeze(0));   // [heads = 1, N]
                algo.m_actions[step] = out.action;                                                       // [N, 1] broadcast into [N, action_size] (:93, :562)
            } else {
                out = algo.m_agent->getActionAndValueDiscrete(next_obs, action);
                algo.m_actions[step] = action.unsqueeze(1);
            }
            algo.m_values[step] = out.value.flatten();
            algo.m_logprobs[step] = out.logprob;
            std::tie(next_obs, reward, done) = algo.stepEnvs(Masked ? out.action.cpu() : action);
            algo.m_rewards[step] = reward.view(-1);
            next_done = done.squeeze();
        }
    }
    g.addI64("crc_obs", { hl::crcOf(algo.m_obs) }, { 1 });
    g.addI64("crc_rewards", { hl::crcOf(algo.m_rewards) }, { 1 });
    g.addI64("crc_dones", { hl::crcOf(algo.m_dones) }, { 1 });
    g.addI64("crc_next_obs", { hl::crcOf(next_obs) }, { 1 });
    g.addI64("crc_next_done", { hl::crcOf(next_done.to(torch::kInt32)) }, { 1 });
    {
        const torch::Tensor dones_sum = algo.m_dones.sum(), next_done_sum = next_done.to(torch::kFloat64).sum();
        g.addF64("count_done", { dones_sum.item<double>(), next_done_sum.item<double>() }, { 2 });
    }
    auto sample = [&](const torch::Tensor& t) { return t.reshape(-1).index({ torch::indexing::Slice(0, torch::indexing::None, 4099) }).clone(); };
    auto sums = [&](const torch::Tensor& t) { torch::Tensor d = t.to(torch::kFloat64); return std::vector<double>{ d.sum().item<double>(), (d * d).sum().item<double>() }; };
    g.add("sample_logprobs", sample(algo.m_logprobs));
    g.add("sample_values", sample(algo.m_values));
    g.addF64("sums_logprobs", sums(algo.m_logprobs), { 2 });
    g.addF64("sums_values", sums(algo.m_values), { 2 });

    // ---- calcAdvantage on synthetic values: bit-exact target (PPO_Discrete.cpp:274-306) ----
    {
        torch::NoGradGuard ng;
        torch::Tensor real_values = algo.m_values.clone();
        std::vector<float> sv(static_cast<size_t>(T * N));
        for (int64_t i = 0; i < T * N; i++) sv[i] = hl::unit24(hl::mix64(hl::SEED_VAL + static_cast<uint64_t>(i))) * 4.0f - 2.0f;
        algo.m_values = torch::from_blob(sv.data(), { T, N }, torch::kFloat32).clone();
        // critic head: weight 0, bias 0.25 -> getValue(next_obs) is exactly 0.25 whatever the hidden layers hold (params[4], params[5]: criticOutputLayer)
        torch::Tensor w_keep = params[4].detach().clone(), b_keep = params[5].detach().clone();
        params[4].zero_(); params[5].fill_(0.25f);
        auto gae = algo.calcAdvantage(next_obs, next_done);
        g.addI64("crc_syn_returns", { hl::crcOf(gae[0]) }, { 1 });
        g.addI64("crc_syn_advantages", { hl::crcOf(gae[1]) }, { 1 });
        g.add("sample_syn_advantages", sample(gae[1]));
        params[4].copy_(w_keep); params[5].copy_(b_keep);
        algo.m_values = real_values;
    }
    // ---- the real advantages, then the update with injected permutations (PPO_Discrete.cpp:554-648) ----
    auto gae = algo.calcAdvantage(next_obs, next_done);
    torch::Tensor returns = gae[0], advantages = gae[1];
    g.add("sample_advantages", sample(advantages));
    g.addF64("sums_advantages", sums(advantages), { 2 });
    g.addF64("sums_returns", sums(returns), { 2 });
    torch::Tensor b_obs = algo.m_obs.reshape({ B, algo.m_obs_size });
    torch::Tensor b_logprobs = algo.m_logprobs.reshape(-1);
    torch::Tensor b_actions = Masked ? algo.m_actions.reshape({ B, A }) : algo.m_actions.reshape(-1);
    torch::Tensor b_advantages = advantages.reshape(-1), b_returns = returns.reshape(-1), b_values = algo.m_values.reshape(-1);
    torch::Tensor b_masks;
    if constexpr (Masked) b_masks = algo.m_action_masks.reshape({ -1, algo.m_action_masks.sizes().back() });
    std::vector<float>().swap(algo.m_clipfracs);
    std::vector<double> scal;
    for (int64_t epoch = 0; epoch < algo.m_update_epochs; epoch++) {
        std::vector<std::pair<uint64_t, int64_t>> keys(static_cast<size_t>(B));
        for (int64_t i = 0; i < B; i++) keys[i] = { hl::mix64(hl::SEED_PERM + static_cast<uint64_t>(epoch * B + i)), i };
        std::sort(keys.begin(), keys.end());
        std::vector<int64_t> pv(static_cast<size_t>(B));
        for (int64_t i = 0; i < B; i++) pv[i] = keys[i].second;
        torch::Tensor b_inds = torch::from_blob(pv.data(), { B }, torch::kInt64).clone();
        for (int64_t start = 0; start < B; start += MB) {
            torch::Tensor mb = b_inds.index({ torch::indexing::Slice(start, start + MB) });
            MinibatchOut r = minibatchLossBackward<Algo, Masked>(algo, mb, b_obs, b_masks, b_actions, b_logprobs, b_advantages, b_returns, b_values);
            torch::Tensor pg_loss = r.pg_loss, v_loss = r.v_loss, entropy_loss = r.entropy_loss, approx_kl = r.approx_kl, loss = r.loss;
            double total_norm = torch::nn::utils::clip_grad_norm_(params, algo.m_max_grad_norm);
            algo.m_optimizer->step();
            scal.insert(scal.end(), { pg_loss.item<double>(), v_loss.item<double>(), entropy_loss.item<double>(), approx_kl.item<double>(),
                                      static_cast<double>(algo.m_clipfracs.back()), loss.item<double>(), total_norm });
        }
    }
    g.addF64("step_scalars", scal, { static_cast<int64_t>(scal.size() / 7), 7 });
    g.add("params_after", flatParams(params));
    g.addF64("lr", { static_cast<torch::optim::AdamWOptions&>(algo.m_optimizer->param_groups()[0].options()).lr() }, { 1 });
    algo.m_threadPool->stop();
    g.save(outPath);
}

// (12) BASELINE.json configs[4]'s OWN SHAPE -- obs 376, heads [3, 3, 3, 2], 4 x 256 tanh bodies -- on the UNMODIFIED reference classes.  The reference's Agent
// hard-wires 2 x 64 bodies in its constructor (Agent.cpp:25-59; the 256-wide third layer is there as comments, :27,32,45-46), but m_Critic, m_Actor,
// m_actionSpace and m_actionSpaceSum are PUBLIC members (Agent.h:44-50), replace_module is public LibTorch API, ppoLayerInit is a public method
// (Agent.cpp:91-99) and getActionAndValueMasked / getValue (Agent.cpp:107-170) are shape-agnostic.  So: a PPO_MultiDiscrete built by its own constructor,
// its agent's two Sequentials swapped for 376 -> 256 x 4 -> {11 | 1} ones initialised by the reference's own ppoLayerInit with the reference's gains, the
// optimizer rebuilt by the constructor's expression (PPO_MultiDiscrete.cpp:78-80), and then the certified minibatch re-drive (minibatchLossBackward above)
// + clip_grad_norm_ + AdamW on a batch whose every input is a counter hash both sides regenerate (tests/test_gpu_config4_ref.py): observations, masks (at least
// one valid action per head), actions (valid under the mask), rewards, done flags, the permutations.  Old log-probs / values are the agent's own on that batch
// (NoGrad, as the rollout computes them, PPO_MultiDiscrete.cpp:559), advantages / returns are calcAdvantage's (:289-346) on them.
// One file carries the initial parameters in full (2.4 MB, shared by both sizes); per size: what is small in full, what is large as a strided sample +
// binary64 sums, the 7 scalars of every optimizer step, per-tensor norms of the first gradient and of the final parameters.
namespace c4 {
constexpr uint64_t SEED_OBS = 0x5101ull << 32, SEED_MASK = 0x5202ull << 32, SEED_KEEP = 0x5303ull << 32, SEED_START = 0x5404ull << 32, SEED_REW = 0x5505ull << 32,
                   SEED_DONE = 0x5606ull << 32, SEED_NOBS = 0x5707ull << 32, SEED_NDONE = 0x5808ull << 32, SEED_PERM = 0x5909ull << 32;
inline float obsOf(uint64_t seed, uint64_t i) { return (hl::unit24(hl::mix64(seed + i)) * 2.0f - 1.0f) * 1.7320508f; }   // uniform, mean 0, variance 1; one rounding
}  // namespace c4

struct Config4Run {   // what a second run of the same scenario, started one ulp away, contributes to the fixture
    std::vector<double> scal;
    torch::Tensor paramsAfterSample, lastGradSample;
    uint32_t crcParamsBefore = 0;
};
Config4Run goldConfig4(int64_t N, int64_t T, int64_t nmb, int64_t epochs, bool full, const std::string& paramsPath, const std::string& outPath, bool ulpTwin = false,
                       const Config4Run* other = nullptr) {
    const std::vector<int64_t> heads = { 3, 3, 3, 2 };
    const int64_t O = 376, A = 11, H = 4, W = 256;
    RunCfg cfg;
    cfg.obs_size = O; cfg.action_size = A; cfg.max_episode_steps = 200; cfg.seed = 1; cfg.num_envs = N; cfg.num_steps = T; cfg.num_minibatches = nmb;
    cfg.update_epochs = epochs; cfg.ent_coef = 0.01; cfg.gamma = 0.99; cfg.updates = 1;
    enterScratchWithConfig(cfg, "config4");
    auto algoPtr = std::make_unique<PPO_MultiDiscrete>();
    PPO_MultiDiscrete& algo = *algoPtr;
    Agent& agent = *algo.m_agent;
    {   // the swap: same layer names as the constructor's (and its commented third layer's), same gains (Agent.cpp:25-37)
        auto lin = [&](int64_t in, int64_t out, double gain) { return agent.ppoLayerInit(torch::nn::Linear(in, out), gain); };
        const double g2 = sqrt(2);
        torch::nn::Sequential critic({ { "criticInputLayer", lin(O, W, g2) }, { "Tanh1", torch::nn::Tanh() }, { "criticMiddleLayer", lin(W, W, g2) }, { "Tanh2", torch::nn::Tanh() },
                                       { "criticMiddleLayer2", lin(W, W, g2) }, { "Tanh3", torch::nn::Tanh() }, { "criticMiddleLayer3", lin(W, W, g2) }, { "Tanh4", torch::nn::Tanh() },
                                       { "criticOutputLayer", lin(W, 1, 1.0) } });
        torch::nn::Sequential actor({ { "actorInputLayer", lin(O, W, g2) }, { "Tanh1", torch::nn::Tanh() }, { "actorMiddleLayer", lin(W, W, g2) }, { "Tanh2", torch::nn::Tanh() },
                                      { "actorMiddleLayer2", lin(W, W, g2) }, { "Tanh3", torch::nn::Tanh() }, { "actorMiddleLayer3", lin(W, W, g2) }, { "Tanh4", torch::nn::Tanh() },
                                      { "actorOutputLayer", lin(W, A, 0.01) } });
        agent.m_Critic = torch::nn::Sequential(agent.replace_module("m_Critic", critic));
        agent.m_Actor = torch::nn::Sequential(agent.replace_module("m_Actor", actor));
        agent.m_actionSpace = heads;
        agent.m_actionSpaceSum = A;
        algo.m_optimizer = std::make_shared<torch::optim::AdamW>(algo.m_agent->parameters(), torch::optim::AdamWOptions(algo.m_learning_rate).eps(1e-5f));
    }
    std::vector<torch::Tensor> params = agent.parameters();
    if (params.size() != 20) throw std::runtime_error("config4: expected 20 parameter tensors (critic first, then actor)");
    const int64_t B = algo.m_batch_size, MB = algo.m_minibatch_size;
    GoldWriter g;
    {
        std::vector<int64_t> meta = { T, N, O, A, algo.m_num_minibatches, algo.m_update_epochs, algo.m_max_episode_steps, algo.m_seed, 1,
                                      algo.m_anneal_lr, algo.m_use_gae, algo.m_norm_adv, algo.m_clip_vloss, 1, W, 4 };
        g.addI64("meta", meta, { static_cast<int64_t>(meta.size()) });
        g.addI64("heads", heads, { H });
        std::vector<float> hp = { algo.m_learning_rate, algo.m_gamma, algo.m_gae_lambda, algo.m_clip_coef, algo.m_ent_coef, algo.m_vf_coef, algo.m_max_grad_norm };
        g.addF32("hparams", hp, { static_cast<int64_t>(hp.size()) });
        std::vector<int64_t> shapes;
        for (const auto& p : params) { shapes.push_back(p.dim() > 0 ? p.size(0) : 1); shapes.push_back(p.dim() > 1 ? p.size(1) : 1); }
        g.addI64("param_shapes", shapes, { static_cast<int64_t>(params.size()), 2 });
    }
    torch::Tensor params0 = flatParams(params);
    g.addI64("crc_params_before", { hl::crcOf(params0) }, { 1 });
    if (ulpTwin) {
        // THE REFERENCE'S OWN SENSITIVITY: the same scenario started from parameters that differ from params0 in the LAST BIT of every element (up or down by
        // a counter hash) -- a change of the size of one fp32 rounding.  The loss has kinks (value clipping, the ratio clip, max(unclipped, clipped)), so two
        // correct fp32 trajectories a rounding apart separate visibly once the first samples reach a kink; the twin's distance from the fixture proper, per step,
        // is the yardstick the device's distance is read against (tests/test_gpu_config4_ref.py).  Thread count is NOT such a yardstick: 1 thread against 8
        // moved the 40 steps' losses by 9e-8 (LibTorch's products do not change their summation order with it).
        torch::NoGradGuard ng;
        uint64_t e = 0;
        for (auto& p : params) {
            float* d = p.data_ptr<float>();
            for (int64_t i = 0; i < p.numel(); i++, e++) d[i] = std::nextafter(d[i], (hl::mix64((0x5A0Aull << 32) + e) & 1u) ? INFINITY : -INFINITY);
        }
    }
    if (!paramsPath.empty()) {
        GoldWriter pw;
        pw.add("params", params0);
        pw.save(paramsPath);
    }

    // ---- the batch: every input a counter hash (time-major [T, N, ...] like the reference's buffers, flat index i = t * N + n) ----
    std::vector<float> obs(static_cast<size_t>(B * O)), nobs(static_cast<size_t>(N * O)), rew(static_cast<size_t>(B)), don(static_cast<size_t>(B));
    std::vector<int64_t> act(static_cast<size_t>(B * H));
    std::vector<uint8_t> msk(static_cast<size_t>(B * A));
    std::vector<int32_t> ndone(static_cast<size_t>(N));
    for (int64_t i = 0; i < B * O; i++) obs[i] = c4::obsOf(c4::SEED_OBS, static_cast<uint64_t>(i));
    for (int64_t i = 0; i < N * O; i++) nobs[i] = c4::obsOf(c4::SEED_NOBS, static_cast<uint64_t>(i));
    for (int64_t i = 0; i < B; i++) {
        rew[i] = hl::unit24(hl::mix64(c4::SEED_REW + static_cast<uint64_t>(i))) * 2.0f - 1.0f;
        don[i] = (hl::mix64(c4::SEED_DONE + static_cast<uint64_t>(i)) % 100u) == 0u ? 1.0f : 0.0f;
        int64_t off = 0;
        for (int64_t h = 0; h < H; h++) {
            const uint64_t w = static_cast<uint64_t>(heads[h]);
            for (uint64_t a = 0; a < w; a++) msk[i * A + off + a] = (hl::mix64(c4::SEED_MASK + static_cast<uint64_t>(i * A + off) + a) % 100u) >= 35u ? 1 : 0;
            msk[i * A + off + static_cast<int64_t>(hl::mix64(c4::SEED_KEEP + static_cast<uint64_t>(i * H + h)) % w)] = 1;     // >= 1 valid action per head
            uint64_t a = hl::mix64(c4::SEED_START + static_cast<uint64_t>(i * H + h)) % w;                                      // the first valid action from a hashed start
            while (!msk[i * A + off + static_cast<int64_t>(a)]) a = (a + 1) % w;
            act[i * H + h] = static_cast<int64_t>(a);
            off += heads[h];
        }
    }
    for (int64_t n = 0; n < N; n++) ndone[n] = (hl::mix64(c4::SEED_NDONE + static_cast<uint64_t>(n)) % 100u) == 0u ? 1 : 0;
    torch::Tensor b_obs = torch::from_blob(obs.data(), { B, O }, torch::kFloat32);
    torch::Tensor b_masks = torch::from_blob(msk.data(), { B, A }, torch::kUInt8).to(torch::kBool);
    torch::Tensor b_actions = torch::from_blob(act.data(), { B, H }, torch::kInt64);
    torch::Tensor next_obs = torch::from_blob(nobs.data(), { N, O }, torch::kFloat32);
    torch::Tensor next_done = torch::from_blob(ndone.data(), { N }, torch::kInt32);
    g.addI64("crc_obs", { hl::crcOf(b_obs) }, { 1 });
    g.addI64("crc_masks", { hl::crcOf(b_masks.to(torch::kUInt8)) }, { 1 });
    g.addI64("crc_actions", { hl::crcOf(b_actions.to(torch::kInt32)) }, { 1 });
    g.addI64("crc_rewards_dones", { hl::crcOf(torch::from_blob(rew.data(), { B }, torch::kFloat32)), hl::crcOf(torch::from_blob(don.data(), { B }, torch::kFloat32)) }, { 2 });

    // ---- the agent's own log-probs / entropies / values on the batch (the rollout's NoGrad call, PPO_MultiDiscrete.cpp:559 -> Agent.cpp:137-170) ----
    AgentOutput roll;
    {
        torch::NoGradGuard ng;
        roll = agent.getActionAndValueMasked(b_obs, b_masks, b_actions.t());
        if (!torch::equal(roll.action, b_actions)) throw std::runtime_error("config4: teacher-forced actions did not come back");
    }
    algo.m_rewards = torch::from_blob(rew.data(), { T, N }, torch::kFloat32).clone();
    algo.m_dones = torch::from_blob(don.data(), { T, N }, torch::kFloat32).clone();
    algo.m_values = roll.value.reshape({ T, N }).clone();
    algo.m_logprobs = roll.logprob.reshape({ T, N }).clone();
    auto gae = algo.calcAdvantage(next_obs, next_done);
    torch::Tensor returns = gae[0], advantages = gae[1];
    torch::Tensor next_value;
    {
        torch::NoGradGuard ng;
        next_value = agent.getValue(next_obs).reshape(-1);
    }
    const int64_t stride = full ? 1 : 4099;
    auto sample = [&](const torch::Tensor& t, int64_t st) { return t.detach().reshape(-1).index({ torch::indexing::Slice(0, torch::indexing::None, st) }).clone(); };
    auto sums = [&](const torch::Tensor& t) { torch::Tensor d = t.detach().to(torch::kFloat64); return std::vector<double>{ d.sum().item<double>(), (d * d).sum().item<double>() }; };
    g.addI64("sample_stride", { stride, 61 }, { 2 });
    for (auto& kv : std::vector<std::pair<std::string, torch::Tensor>>{ { "logprobs", roll.logprob }, { "entropy", roll.entropy }, { "values", roll.value }, { "advantages", advantages },
                                                                        { "returns", returns }, { "next_value", next_value } }) {
        g.add("sample_" + kv.first, sample(kv.second, kv.first == "next_value" ? 1 : stride));
        g.addF64("sums_" + kv.first, sums(kv.second), { 2 });
    }

    // ---- the update: injected permutations, the certified minibatch expressions, clip_grad_norm_, AdamW (PPO_MultiDiscrete.cpp:593-668) ----
    torch::Tensor b_logprobs = algo.m_logprobs.reshape(-1), b_advantages = advantages.reshape(-1), b_returns = returns.reshape(-1), b_values = algo.m_values.reshape(-1);
    std::vector<float>().swap(algo.m_clipfracs);
    Config4Run run;
    run.crcParamsBefore = hl::crcOf(params0);
    if (other && other->crcParamsBefore != run.crcParamsBefore) throw std::runtime_error("config4: the two runs did not start from the same parameters");
    std::vector<double>& scal = run.scal;
    auto tensorNorms = [&](bool grads) { std::vector<double> v; for (const auto& p : params) v.push_back((grads ? p.grad() : p).detach().to(torch::kFloat64).norm().item<double>()); return v; };
    int64_t k = 0;
    const int64_t steps = algo.m_update_epochs * ((B + MB - 1) / MB);
    for (int64_t epoch = 0; epoch < algo.m_update_epochs; epoch++) {
        std::vector<std::pair<uint64_t, int64_t>> keys(static_cast<size_t>(B));
        for (int64_t i = 0; i < B; i++) keys[i] = { hl::mix64(c4::SEED_PERM + static_cast<uint64_t>(epoch * B + i)), i };
        std::sort(keys.begin(), keys.end());
        std::vector<int64_t> pv(static_cast<size_t>(B));
        for (int64_t i = 0; i < B; i++) pv[i] = keys[i].second;
        torch::Tensor b_inds = torch::from_blob(pv.data(), { B }, torch::kInt64).clone();
        for (int64_t start = 0; start < B; start += MB) {
            torch::Tensor mb = b_inds.index({ torch::indexing::Slice(start, start + MB) });
            MinibatchOut r = minibatchLossBackward<PPO_MultiDiscrete, true>(algo, mb, b_obs, b_masks, b_actions, b_logprobs, b_advantages, b_returns, b_values);
            if (k == 0 || k == steps - 1) {
                const std::string K = "k" + std::to_string(k) + "/";
                g.addF64(K + "grad_norms", tensorNorms(true), { 20 });
                g.add(K + "sample_grads", sample(flatParams(params, true), 61));
                if (k == steps - 1) run.lastGradSample = sample(flatParams(params, true), 61);
                g.add(K + "sample_newlogprob", sample(r.o.logprob, full ? 1 : 509));
                g.add(K + "sample_newvalue", sample(r.o.value, full ? 1 : 509));
                g.add(K + "sample_entropy", sample(r.o.entropy, full ? 1 : 509));
            }
            double total_norm = torch::nn::utils::clip_grad_norm_(params, algo.m_max_grad_norm);
            algo.m_optimizer->step();
            if (k == 0) g.add("k0/sample_params_after", sample(flatParams(params), 61));
            scal.insert(scal.end(), { r.pg_loss.item<double>(), r.v_loss.item<double>(), r.entropy_loss.item<double>(), r.approx_kl.item<double>(),
                                      static_cast<double>(algo.m_clipfracs.back()), r.loss.item<double>(), total_norm });
            std::cerr << "[ref_harness] config4 " << N << "x" << T << " step " << k << ": loss " << r.loss.item<double>() << " pg " << r.pg_loss.item<double>() << " v " << r.v_loss.item<double>()
                      << " kl " << r.approx_kl.item<double>() << " norm " << total_norm << "\n";
            k++;
        }
    }
    g.addF64("step_scalars", scal, { k, 7 });
    torch::Tensor paramsAfter = flatParams(params);
    g.add("sample_params_after", sample(paramsAfter, 61));
    g.addF64("sums_params_after", sums(paramsAfter), { 2 });
    g.addF64("norms_params_after", tensorNorms(false), { 20 });
    g.addF64("max_abs_param_change", { (paramsAfter - params0).abs().max().item<double>() }, { 1 });
    g.addF64("lr", { static_cast<torch::optim::AdamWOptions&>(algo.m_optimizer->param_groups()[0].options()).lr() }, { 1 });
    run.paramsAfterSample = sample(paramsAfter, 61);
    if (other) {
        // the same scenario run by the same unmodified reference from parameters one ulp away (ulpTwin above)
        g.addF64("ulp_twin/step_scalars", other->scal, { static_cast<int64_t>(other->scal.size() / 7), 7 });
        g.add("ulp_twin/sample_params_after", other->paramsAfterSample);
        g.add("ulp_twin/sample_last_grads", other->lastGradSample);
    }
    g.addI64("torch_threads", { at::get_num_threads() }, { 1 });
    if (!outPath.empty()) g.save(outPath);
    return run;
}

// (10) Learning curves: the reference's own acceptance test is "run ./PPO and watch ep_len_mean" (README.md:169-178).  The UNMODIFIED train()
// (PPO_Discrete.cpp:485-690) runs to total_timesteps with the recommended hyper-parameters; the table it prints per update
// (printPPOResults, :700-774) is captured and parsed into one JSON object per seed: exactly the numbers a user of the reference sees
// (ep_len_mean with the table's 2 decimals).  tests/golden/curves_*.json = these objects for several seeds (oracle/make_curves.py).
template <class Algo>
int curvesReference(const RunCfg& c, const std::string& outFile) {
    enterScratchWithConfig(c, "curves");
    std::stringstream ss;
    std::cout.copyfmt(std::ios(nullptr));
    std::streambuf* old = std::cout.rdbuf(ss.rdbuf());
    try { Algo algo; algo.train(); } catch (...) { std::cout.rdbuf(old); throw; }
    std::cout.rdbuf(old);
    const char* keys[] = { "ep_len_mean", "ep_rew_mean", "total_timesteps", "approx_kl", "clip_fraction", "explained_variance", "learning_rate",
                           "loss", "policy_gradient_loss", "value_loss", "entropy_loss" };
    constexpr int NK = sizeof keys / sizeof keys[0];
    std::vector<std::array<std::string, NK>> rows;   // one per printed table; "null" where the table has no such row (first update, no finished episode)
    std::array<std::string, NK> cur; cur.fill("null");
    bool open = false;
    std::istringstream is(ss.str());
    std::string l;
    while (std::getline(is, l)) {
        if (!l.empty() && l[0] == '-') {   // a rule line opens or closes a table
            if (open) { rows.push_back(cur); cur.fill("null"); }
            open = !open;
            continue;
        }
        if (!open || l.size() < 6 || l[0] != '|') continue;
        const size_t bar = l.find('|', 1);
        if (bar == std::string::npos) continue;
        std::string k = l.substr(1, bar - 1), v = l.substr(bar + 1);
        auto trim = [](std::string s) { size_t a = s.find_first_not_of(" |"), b = s.find_last_not_of(" |"); return a == std::string::npos ? std::string() : s.substr(a, b - a + 1); };
        k = trim(k); v = trim(v);
        for (int i = 0; i < NK; i++) if (k == keys[i] && !v.empty()) cur[i] = (v == "nan" || v == "-nan" || v == "inf") ? "null" : v;
    }
    std::ofstream f(outFile, std::ios::binary);
    f << "{\"seed\": " << c.seed << ", \"updates\": " << rows.size();
    for (int i = 0; i < NK; i++) {
        f << ", \"" << keys[i] << "\": [";
        for (size_t r = 0; r < rows.size(); r++) f << (r ? ", " : "") << rows[r][i];
        f << "]";
    }
    f << "}\n";
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    try {
        std::string mode = argc > 1 ? argv[1] : "";
        if (mode == "golden" && argc > 2) {
            char buf[4096];
            std::string out = argv[2];
            if (out[0] != '/') out = std::string(getcwd(buf, sizeof buf)) + "/" + out;
            goldResetStream(out + "/cartpole_reset_stream.pgld");
            goldCartPoleTransitions(out + "/cartpole_transitions.pgld");
            goldMountainCarTransitions(out + "/mountaincar_transitions.pgld");
            goldDistributions(out + "/distributions.pgld");
            goldMultiHeadAgent(out + "/multihead_agent.pgld");
            {
                RunCfg c;  // shipped-TOML shape with action_size = 2; two updates cover the LR anneal
                c.updates = 2;
                goldTrainScenario<PPO_Discrete, false>(c, "t32n8", out + "/discrete_t32_n8_seed2.pgld");
            }
            {
                RunCfg c;  // truncation at max_episode_steps inside a short rollout; no adv-norm / value clipping / anneal
                c.seed = 3; c.num_envs = 16; c.num_steps = 64; c.max_episode_steps = 20; c.update_epochs = 2;
                c.norm_adv = false; c.clip_vloss = false; c.anneal_lr = false; c.ent_coef = 0.01; c.gamma = 0.99;
                goldTrainScenario<PPO_Discrete, false>(c, "t64n16", out + "/discrete_t64_n16_seed3_trunc.pgld");
            }
            {
                RunCfg c;
                c.seed = 1; c.num_envs = 64; c.num_steps = 128; c.update_epochs = 2; c.ent_coef = 0.01;
                goldTrainScenario<PPO_Discrete, false>(c, "t128n64", out + "/discrete_t128_n64_seed1.pgld");
            }
            {
                RunCfg c;  // PPO_MultiDiscrete + MountainCar (masked categorical, true entropy)
                c.obs_size = 2; c.action_size = 3; c.max_episode_steps = 200; c.seed = 1; c.num_envs = 16; c.num_steps = 32;
                c.update_epochs = 3; c.ent_coef = 0.01; c.gamma = 0.99;
                goldTrainScenario<PPO_MultiDiscrete, true>(c, "mc", out + "/multidiscrete_mountaincar_t32_n16.pgld");
            }
            return 0;
        }
        if (mode == "golden_shipped" && argc > 2) {
            // round 3 (missing #3): the shapes the reference itself ships and BASELINE.json configs[0] names, as they are
            char buf[4096];
            std::string out = argv[2];
            if (out[0] != '/') out = std::string(getcwd(buf, sizeof buf)) + "/" + out;
            {
                RunCfg c;  // Environments/CartPoleRecommendedSettings.toml AS SHIPPED: action_size = 1 (always action 0), 8 envs x 32 steps, seed 2
                c.action_size = 1; c.updates = 2;
                goldTrainScenario<PPO_Discrete, false>(c, "shipped", out + "/discrete_shipped_toml_t32_n8_act1.pgld");
            }
            {
                RunCfg c;  // BASELINE.json configs[0]: 8 parallel envs, 128 steps, the recommended settings with action_size = 2
                c.num_steps = 128; c.updates = 2;
                goldTrainScenario<PPO_Discrete, false>(c, "config0", out + "/discrete_config0_t128_n8_seed2.pgld");
            }
            return 0;
        }
        if (mode == "hostgold" && argc > 2) {
            char buf[4096];
            std::string out = argv[2];
            if (out[0] != '/') out = std::string(getcwd(buf, sizeof buf)) + "/" + out;
            goldHost(out);
            goldGetArgs(out);
            goldTrainRun<PPO_Discrete>(out + "/train_run.txt",
                "[environment]\nobs_size = 4\naction_size = 2\nmax_episode_steps = 500\n\n"
                "[general]\nseed = 5\ntotal_timesteps = 384\nuse_cuda = false\ntorch_deterministic = true\ncheckpoint_updates = 2\n\n"
                "[ppo]\nlearning_rate = 0.001\nnum_envs = 8\nnum_steps = 16\nanneal_lr = true\nnum_minibatches = 2\nupdate_epochs = 2\n");
            goldTrainRun<PPO_MultiDiscrete>(out + "/train_run_multidiscrete.txt",
                "[environment]\nobs_size = 2\naction_size = 3\naction_high = 1.0\naction_low = -1.0\nmax_episode_steps = 50\n\n"
                "[general]\nseed = 9\ntotal_timesteps = 256\ncheckpoint_updates = 1\n\n"
                "[ppo]\nlearning_rate = 0.0005\nnum_envs = 4\nnum_steps = 32\nnum_minibatches = 4\nupdate_epochs = 1\nent_coef = 0.01\n");
            return 0;
        }
        if (mode == "ptgold" && argc > 2) {
            char buf[4096];
            std::string out = argv[2];
            if (out[0] != '/') out = std::string(getcwd(buf, sizeof buf)) + "/" + out;
            goldCheckpointFiles<PPO_Discrete>(out, "discrete",
                "[environment]\nobs_size = 4\naction_size = 2\nmax_episode_steps = 500\n\n"
                "[general]\nseed = 5\ntotal_timesteps = 384\nuse_cuda = false\ntorch_deterministic = true\ncheckpoint_updates = 2\n\n"
                "[ppo]\nlearning_rate = 0.001\nnum_envs = 8\nnum_steps = 16\nanneal_lr = true\nnum_minibatches = 2\nupdate_epochs = 2\n", 384);
            goldCheckpointFiles<PPO_MultiDiscrete>(out, "multidiscrete",
                "[environment]\nobs_size = 2\naction_size = 3\naction_high = 1.0\naction_low = -1.0\nmax_episode_steps = 50\n\n"
                "[general]\nseed = 9\ntotal_timesteps = 256\ncheckpoint_updates = 1\n\n"
                "[ppo]\nlearning_rate = 0.0005\nnum_envs = 4\nnum_steps = 32\nnum_minibatches = 4\nupdate_epochs = 1\nent_coef = 0.01\n", 256);
            return 0;
        }
        if (mode == "ptload" && argc > 6) {
            loadCheckpointFiles(argv[2], argv[3], std::atol(argv[4]), std::atol(argv[5]), argv[6]);
            return 0;
        }
        if (mode == "bench" && argc > 4) return benchReference(std::atol(argv[2]), std::atol(argv[3]), std::atol(argv[4]), argc > 5 ? std::atol(argv[5]) : 0);
        if (mode == "headline" && argc > 2) {
            char buf[4096];
            std::string out = argv[2];
            if (out[0] != '/') out = std::string(getcwd(buf, sizeof buf)) + "/" + out;
            {
                RunCfg c;   // BASELINE.json configs[1]: CartPole-v1, 4096 envs x 128 steps, the recommended settings with action_size = 2
                c.num_envs = 4096; c.num_steps = 128; c.seed = 2;
                goldHeadline<PPO_Discrete, false>(c, "cartpole", out + "/headline_cartpole_4096x128.pgld");
            }
            {
                RunCfg c;   // BASELINE.json configs[3]: MountainCar, 8192 envs, CategoricalMasked path
                c.obs_size = 2; c.action_size = 3; c.max_episode_steps = 200; c.seed = 1; c.num_envs = 8192; c.num_steps = 128; c.ent_coef = 0.01; c.gamma = 0.99;
                goldHeadline<PPO_MultiDiscrete, true>(c, "mountaincar", out + "/headline_mountaincar_8192x128.pgld");
            }
            return 0;
        }
        if (mode == "config4" && argc > 2) {
            char buf[4096];
            std::string out = argv[2];
            if (out[0] != '/') out = std::string(getcwd(buf, sizeof buf)) + "/" + out;
            // optional: LibTorch intra-op threads.  MKL / ATen pick their summation order by thread count, so a second run with another count is the REFERENCE'S
            // OWN fp32 noise over the 40 steps -- the yardstick for how far two correct fp32 trajectories drift apart (tools/c4_ref_self_distance.py)
            // a size the scalar oracle can follow, with every per-sample tensor in full (64 envs x 16 steps, 4 minibatches of 256, 2 epochs) ...
            Config4Run small1 = goldConfig4(64, 16, 4, 2, true, "", "", true);
            goldConfig4(64, 16, 4, 2, true, out + "/config4_params.pgld", out + "/config4_small_64x16.pgld", false, &small1);
            // ... and configs[4]'s per-GPU share: 16 384 envs / 8 GPUs = 2048 envs x 128 steps, 4 minibatches of 65 536 rows, the recommended 10 epochs;
            // first the twin one ulp away (kept as "ulp_twin/*": the reference's own sensitivity), then the fixture proper
            Config4Run share1 = goldConfig4(2048, 128, 4, 10, false, "", "", true);
            goldConfig4(2048, 128, 4, 10, false, "", out + "/config4_share_2048x128.pgld", false, &share1);
            return 0;
        }
        if (mode == "curves" && argc > 6) {
            // CartPoleRecommendedSettings.toml's hyper-parameters (RunCfg's defaults) with action_size = 2, as BASELINE.json configs[0] runs them
            char buf[4096];
            std::string out = argv[2];
            if (out[0] != '/') out = std::string(getcwd(buf, sizeof buf)) + "/" + out;
            RunCfg c;
            c.num_envs = std::atol(argv[3]); c.num_steps = std::atol(argv[4]); c.total_timesteps = std::atol(argv[5]); c.seed = std::atol(argv[6]);
            return curvesReference<PPO_Discrete>(c, out);
        }
        std::cerr << "usage: ref_harness golden <outdir> | hostgold <outdir> | ptgold <outdir> | ptload <agent.pt> <optimizer.pt> <obs> <act> <out.pgld> | "
                     "bench <num_envs> <num_steps> <updates> [threads] | curves <out.json> <num_envs> <num_steps> <total_timesteps> <seed> | headline <outdir> | config4 <outdir>\n";
        return 2;
    } catch (const std::exception& ex) {
        std::cerr << "[ref_harness] error: " << ex.what() << std::endl;
        return 1;
    }
}
