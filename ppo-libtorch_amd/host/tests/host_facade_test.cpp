// Exercises the C++ classes with the reference's names on a real GPU (run by tests/test_host_facade.py, -m gpu).
// Prints "HOST_FACADE_OK" when every check holds.
#include <atomic>
#include <algorithm>
#include <map>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iomanip>
#include <sstream>

#include "../PPO/PPO_Discrete.h"
#include "../PPO/PPO_MultiDiscrete.h"
#include "../Utils/TorchArchive.h"

#define REQUIRE(cond)                                                                 \
    do {                                                                              \
        if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
    } while (0)

static void writeConfig(const char* body) { std::ofstream("PPOConfig.toml") << body; }

// Splits a fixture line "a|b|c" at the bars.
static std::vector<std::string> bars(const std::string& line) {
    std::vector<std::string> out;
    size_t at = 0;
    for (;;) {
        const size_t n = line.find('|', at);
        out.push_back(line.substr(at, n == std::string::npos ? n : n - at));
        if (n == std::string::npos) break;
        at = n + 1;
    }
    return out;
}

// argv[1]: tests/golden (the fixtures oracle/ref_harness wrote from the reference itself)
int main(int argc, char** argv) {
    namespace fs = std::filesystem;
    const std::string golden = argc > 1 ? fs::absolute(argv[1]).string() : "";
    const std::string keep = argc > 2 ? fs::absolute(argv[2]).string() : "";   // where the final checkpoint pair of the training run below is left for the caller
    const fs::path scratch = fs::temp_directory_path() / "ppo_host_facade_test";
    fs::remove_all(scratch);
    fs::create_directories(scratch);
    fs::current_path(scratch);
    auto dev = std::make_shared<ppo::Device>(0);

    // ---- environments: the reference's duck type
    CartPole env(2, dev);
    std::vector<float> s0 = env.reset();
    REQUIRE(s0.size() == 4 && std::fabs(s0[0] - (-0.00640051067f)) < 1e-9f);   // first draw of mt19937(2), SURVEY 8(a) a2
    auto [obs, rew, term, info] = env.step(1);
    REQUIRE(obs.size() == 4 && rew == 1.0f && !term && !info && env.episode_length == 1 && env.episode_reward == 1.0f);
    REQUIRE(obs[0] == s0[0] + 0.02f * s0[1]);                                  // positions move with the OLD velocities (CartPole.cpp:66)
    MountainCar car(dev, 1, 0);
    auto cs = car.reset();
    REQUIRE(cs[0] >= -0.6f && cs[0] <= -0.4f && cs[1] == 0.0f);
    auto [cobs, crew, cterm, cinfo] = car.step(2);
    REQUIRE(crew == -1.0f && !cterm && cobs.size() == 2);
    REQUIRE(car.getActionMask().cpu<uint8_t>() == std::vector<uint8_t>({ 1, 1, 1 }));

    // ---- distributions
    ppo::Tensor logits = ppo::Tensor::from_host<float>(dev, { 0.0f, 0.0f, 1.0f, -1.0f }, { 2, 2 });
    Categorical cat(logits, dev);
    auto probs = cat.m_probs.cpu<float>();
    REQUIRE(std::fabs(probs[0] - 0.5f) < 1e-6f && std::fabs(probs[2] - 0.880797f) < 1e-5f);
    auto lp = cat.log_prob(ppo::Tensor::from_host<int64_t>(dev, { 1, 0 }, { 2 })).cpu<float>();
    REQUIRE(std::fabs(lp[0] - std::log(0.5f)) < 1e-6f && std::fabs(lp[1] - std::log(0.880797f)) < 1e-5f);
    REQUIRE(std::fabs(cat.entropy().cpu<float>()[0]) < 2e-38f);               // the reference's clamp bug (Categorical.cpp:112-119)
    REQUIRE(cat.mode().cpu<int64_t>()[1] == 0 && cat.sample().numel() == 2);
    CategoricalMasked cm(logits, ppo::Tensor::from_host<uint8_t>(dev, { 1, 0, 1, 1 }, { 2, 2 }), dev);
    REQUIRE(std::fabs(cm.m_probs.cpu<float>()[0] - 1.0f) < 1e-6f && cm.entropy().cpu<float>()[1] > 0.3f);

    // ---- agent
    Agent agent(4, 2, dev);
    REQUIRE(agent.parameterCount() == 9155 && agent.m_actionSpaceSum == 2);
    ppo::Tensor x = ppo::Tensor::from_host<float>(dev, { 0.01f, -0.02f, 0.03f, 0.04f, 0.0f, 0.0f, 0.0f, 0.0f }, { 2, 4 });
    AgentOutput out = agent.getActionAndValueDiscrete(x);
    REQUIRE(out.action.numel() == 2 && out.value.numel() == 2);
    AgentOutput again = agent.getActionAndValueDiscrete(x, out.action);
    REQUIRE(again.logprob.cpu<float>() == out.logprob.cpu<float>());
    REQUIRE(agent.getValue(x).cpu<float>() == out.value.cpu<float>());

    // ---- algorithm: config keys, obs-size error, a short training run, checkpoint / resume
    writeConfig("[environment]\nobs_size = 3\naction_size = 2\n[general]\nseed = 2\n[ppo]\nnum_envs = 8\nnum_steps = 16\n");
    bool threw = false;
    try { PPO_Discrete bad; } catch (const std::runtime_error& e) {
        threw = std::string(e.what()).find("The environment returned an observation of size 4, but your config defined") != std::string::npos;
    }
    REQUIRE(threw);
    writeConfig("# shipped CartPoleRecommendedSettings.toml with action_size = 2 and a short run\n"
                "[environment]\nobs_size = 4\naction_size = 2\nmax_episode_steps = 500\n"
                "[general]\nseed = 2\ntotal_timesteps = 32768\nuse_cuda = true\ntorch_deterministic = true\ncheckpoint_updates = 4\n"
                "[ppo]\nlearning_rate = 0.001\nnum_envs = 64\nnum_steps = 64\nanneal_lr = true\nuse_gae = true\ngamma = 0.98\ngae_lambda = 0.95\n"
                "num_minibatches = 4\nupdate_epochs = 4\nnorm_adv = true\nclip_coef = 0.2\nclip_vloss = true\nent_coef = 0.0\nvf_coef = 0.5\nmax_grad_norm = 0.5\n");
    {
        PPO_Discrete algo;
        REQUIRE(algo.m_num_envs == 64 && algo.m_batch_size == 4096 && algo.m_minibatch_size == 1024 && algo.m_gamma == 0.98f);
        ppo::Tensor first = algo.initEnvs();
        REQUIRE(first.numel() == 64 * 4);
        AgentOutput a = algo.computeActionLogic(first);
        auto [o2, r2, d2] = algo.stepEnvs(a.action);
        REQUIRE(o2.numel() == 256 && r2.cpu<float>()[0] == 1.0f && d2.cpu<int32_t>()[0] == 0);
        auto adv = algo.calcAdvantage(ppo::Tensor(), ppo::Tensor());
        REQUIRE(adv[0].numel() == 4096 && adv[1].numel() == 4096);
        algo.train();
        REQUIRE(algo.m_global_step == 32768);
        REQUIRE(fs::exists("./Models/PPO_Agent_32768_steps.pt") && fs::exists("./ModelCheckpoints/PPO_Agent_16384_steps.pt") &&
                fs::exists("./OptimizerCheckpoints/PPO_Optimizer_32768_steps.pt"));
        REQUIRE(!algo.m_episode_stats->empty() && algo.m_episode_stats->avgLength() > 8.0);
        if (!keep.empty()) {   // tests/test_host_facade.py hands these to the compiled reference's own torch::load calls
            fs::create_directories(keep);
            fs::copy_file("./Models/PPO_Agent_32768_steps.pt", keep + "/PPO_Agent_32768_steps.pt", fs::copy_options::overwrite_existing);
            fs::copy_file("./Models/PPO_Optimizer_32768_steps.pt", keep + "/PPO_Optimizer_32768_steps.pt", fs::copy_options::overwrite_existing);
        }
    }
    {
        PPO_Discrete resumed;   // picks up the newest checkpoint (mtime) and its step count from the file name
        REQUIRE(resumed.m_global_step == 32768);
        // ... and holds what the files hold: parameters, AdamW moments, step count (8 updates x 4 epochs x 4 minibatches)
        const ppo::pt::AgentFile fa = ppo::pt::readAgent("./ModelCheckpoints/PPO_Agent_32768_steps.pt");
        const ppo::pt::OptimizerFile fo = ppo::pt::readOptimizer("./OptimizerCheckpoints/PPO_Optimizer_32768_steps.pt");
        const int64_t P = ppo_param_count(resumed.m_ctx);
        std::vector<float> p((size_t)P), m((size_t)P), v((size_t)P), want_p, want_m;
        int64_t step = 0;
        REQUIRE(ppo_params_get_h(resumed.m_ctx, p.data(), P) == PPO_OK && ppo_optimizer_get_h(resumed.m_ctx, m.data(), v.data(), P, &step) == PPO_OK);
        for (const auto& t : fa.tensors) want_p.insert(want_p.end(), t.values.begin(), t.values.end());
        for (const auto& t : fo.exp_avg) want_m.insert(want_m.end(), t.values.begin(), t.values.end());
        REQUIRE(fa.tensors.size() == 12 && fa.tensors[0].name == "m_Critic.criticInputLayer.weight" && fa.tensors[11].name == "m_Actor.actorOutputLayer.bias");
        REQUIRE(want_p.size() == (size_t)P && std::memcmp(want_p.data(), p.data(), (size_t)P * 4) == 0);
        REQUIRE(want_m.size() == (size_t)P && std::memcmp(want_m.data(), m.data(), (size_t)P * 4) == 0);
        REQUIRE(step == 8 * 4 * 4 && fo.step.size() == 12 && fo.step[0] == step && fo.eps == (double)1e-5f);
    }
    fs::remove_all("./ModelCheckpoints"); fs::remove_all("./OptimizerCheckpoints");
    writeConfig("[environment]\nobs_size = 2\naction_size = 3\nmax_episode_steps = 200\n[general]\nseed = 1\ntotal_timesteps = 8192\ncheckpoint_updates = 100\n"
                "[ppo]\nnum_envs = 32\nnum_steps = 64\nnum_minibatches = 4\nupdate_epochs = 2\n");
    {
        PPO_MultiDiscrete md;
        ppo::Tensor mask(dev, { 32, 3 }, ppo::DType::u8);
        ppo::Tensor first = md.initEnvs(mask);
        REQUIRE(mask.cpu<uint8_t>()[5] == 1);
        AgentOutput a = md.computeActionLogic(first, mask);
        REQUIRE(a.entropy.cpu<float>()[0] > 1.0f);   // ~ln 3: true entropy on the masked path
        md.train();
        REQUIRE(md.m_global_step == 8192);
    }
    // ---- ThreadPool (Utils/ThreadPool.cpp:19-139): every queued job runs once, waitForJobsToFinish returns only when the queue is empty and no
    //      job is running, the pool can be restarted, an exception inside a job is swallowed (:43-49) and does not kill its worker
    {
        ThreadPool pool(4);
        pool.start();
        std::atomic<int> ran{ 0 };
        for (int round = 0; round < 3; round++) {
            for (int i = 0; i < 500; i++) pool.queueJob([&ran, i] { if (i == 250) throw std::runtime_error("job failure"); ran.fetch_add(1); });
            pool.waitForJobsToFinish();
            REQUIRE(ran.load() == 499 * (round + 1) && !pool.busy());
        }
        pool.stop();
        pool.start();
        pool.queueJob([&ran] { ran.fetch_add(1); });
        pool.waitForJobsToFinish();
        REQUIRE(ran.load() == 3 * 499 + 1);
        pool.stop();
    }
    // ---- use_cuda = false (the reference's shipped CartPoleRecommendedSettings.toml): a warning, not an exception; a checkpoint directory whose
    //      newest file is no readable archive (cut short; another program's): reported with the reason and ignored, the agent starts fresh
    fs::remove_all("./ModelCheckpoints"); fs::remove_all("./OptimizerCheckpoints"); fs::remove_all("./Models");
    fs::create_directories("./ModelCheckpoints"); fs::create_directories("./OptimizerCheckpoints");
    std::ofstream("./ModelCheckpoints/PPO_Agent_4242_steps.pt") << "PK\x03\x04 the start of an archive and nothing else";
    std::ofstream("./OptimizerCheckpoints/PPO_Optimizer_4242_steps.pt") << "PPOHIP01 some other program's file";
    writeConfig("[environment]\nobs_size = 4\naction_size = 2\n[general]\nseed = 2\ntotal_timesteps = 2560\nuse_cuda = false\n"
                "[ppo]\nnum_envs = 8\nnum_steps = 32\nupdate_epochs = 10\n");
    {
        std::stringstream captured;
        std::streambuf* old = std::cout.rdbuf(captured.rdbuf());
        PPO_Discrete algo;
        std::cout.rdbuf(old);
        REQUIRE(captured.str().find("use_cuda = false requested") != std::string::npos);
        REQUIRE(captured.str().find("PPO_Agent_4242_steps.pt: has no ZIP end-of-central-directory record (truncated?); ignoring it.") != std::string::npos);
        REQUIRE(captured.str().find("PPO_Optimizer_4242_steps.pt: has no ZIP end-of-central-directory record (truncated?); ignoring it.") != std::string::npos);
        REQUIRE(algo.m_global_step == 0 && algo.m_batch_size == 256);
        // ---- the console table (PPO_Discrete.cpp:700-774) against the reference's own printout of the same four calls (tests/golden/console_table.txt,
        //      written by oracle/ref_harness hostgold): byte for byte, manipulator state carried from call to call as in train()
        if (!golden.empty()) {
            std::ifstream gf(golden + "/console_table.txt", std::ios::binary);
            REQUIRE(gf.good());
            std::stringstream want; want << gf.rdbuf();
            REQUIRE(ppo_set_learning_rate(algo.m_ctx, (double)0.00075f) == PPO_OK);
            algo.m_clipfracs = { 0.125f, 0.0625f, 0.25f };
            auto scalar = [&](float v) { return ppo::Tensor::from_host<float>(dev, { v }, { 1 }); };
            ppo::Tensor kl = scalar(0.00123456789f), ent = scalar(-1.17549435e-38f), ev = scalar(0.1762397289f), loss = scalar(24.916658401f),
                        pg = scalar(-0.007237161f), vl = scalar(27.17522430f);
            std::stringstream got;
            old = std::cout.rdbuf(got.rdbuf());
            algo.m_episode_stats = std::make_unique<CircularBuffer>(100);
            algo.m_episode_stats->add(21.0f, 22); algo.m_episode_stats->add(13.0f, 14); algo.m_episode_stats->add(-1.0f, 1); algo.m_episode_stats->add(499.0f, 500);
            algo.printPPOResults(1, 256, std::chrono::milliseconds(123), std::chrono::milliseconds(4567), kl, ent, ev, loss, pg, vl);
            algo.printPPOResults(2, 512, std::chrono::milliseconds(97), std::chrono::milliseconds(12345), kl, ent, ev, loss, pg, vl);
            algo.m_episode_stats = std::make_unique<CircularBuffer>(100);
            algo.printPPOResults(3, 768, std::chrono::milliseconds(97), std::chrono::milliseconds(23456), kl, ent, ev, loss, pg, vl);
            algo.printPPOResults(1, 256, std::chrono::milliseconds(123), std::chrono::milliseconds(999), kl, ent, ev, loss, pg, vl);
            std::cout.rdbuf(old);
            if (got.str() != want.str()) {
                std::fprintf(stderr, "console table differs from the reference's:\n--- got\n%s\n--- want\n%s\n", got.str().c_str(), want.str().c_str());
                return 1;
            }
        }
    }
    // ---- PPOUtils (Utils.cpp:5-60) on the reference's own answers (tests/golden/host_utils.txt)
    if (!golden.empty()) {
        std::ifstream uf(golden + "/host_utils.txt");
        REQUIRE(uf.good());
        std::string line;
        int checked = 0;
        while (std::getline(uf, line)) {
            const std::vector<std::string> f = bars(line);
            if (f[0] == "steps") {
                REQUIRE(f.size() == 5);
                const std::string r = PPOUtils::getLoadFromSteps(f[1], f[2]);
                REQUIRE(r == f[3] && PPOUtils::isNumber(r) == (f[4] == "1"));
            } else if (f[0] == "isnum") {
                REQUIRE(f.size() == 3 && PPOUtils::isNumber(f[1]) == (f[2] == "1"));
            } else if (f[0] == "mean") {
                REQUIRE(f.size() == 3);
                std::vector<float> v;
                std::stringstream ss(f[1]);
                std::string tok;
                while (std::getline(ss, tok, ',')) { const uint32_t b = (uint32_t)std::stoul(tok, nullptr, 16); float x; std::memcpy(&x, &b, 4); v.push_back(x); }
                const float m = PPOUtils::getVectorMean(v);
                uint32_t mb; std::memcpy(&mb, &m, 4);
                REQUIRE(mb == (uint32_t)std::stoul(f[2], nullptr, 16));
            }
            checked++;
        }
        REQUIRE(checked >= 16);
    }
    // ---- the constructor's console output and the hyper-parameters it ends up with (PPO_Discrete.cpp:10-105, getArgs :107-255) against the
    //      reference's own, for a full PPOConfig.toml, a partial one and none (tests/golden/getargs.txt carries the TOML text, the reference's
    //      stdout and its fields).  The device line is this build's own, and so is the use_cuda = false warning.
    if (!golden.empty()) {
        struct ArgsProbe : PPOAlgorithm {   // getArgs without the GPU half of the constructor (the defaults describe no runnable env: obs 2, 1 action)
            explicit ArgsProbe(bool multi) : PPOAlgorithm(multi ? PPO_ENV_MOUNTAINCAR : PPO_ENV_CARTPOLE, multi ? PPO_DIST_MASKED : PPO_DIST_CATEGORICAL, 2, multi ? 200 : 500) { getArgs(); }
        };
        std::ifstream gf(golden + "/getargs.txt", std::ios::binary);
        REQUIRE(gf.good());
        std::string line, name, section, toml, out;
        std::vector<std::pair<std::string, std::string>> fields;
        int variants = 0;
        auto is_device_line = [](const std::string& l) { return l.rfind("Using ", 0) == 0 && l.find(" device") != std::string::npos; };
        auto check = [&]() -> int {
            if (name.empty()) return 0;
            const fs::path dir = scratch / ("getargs_" + name);
            fs::create_directories(dir);
            fs::current_path(dir);
            if (toml != "(none)\n") std::ofstream("PPOConfig.toml", std::ios::binary) << toml;
            std::vector<std::string> want_args, want_rest;   // the reference's lines before / after its device line
            {
                std::istringstream is(out);
                std::string l; bool after = false;
                while (std::getline(is, l)) { if (is_device_line(l)) { after = true; continue; } (after ? want_rest : want_args).push_back(l); }
            }
            std::stringstream ss;
            std::cout.copyfmt(std::ios(nullptr));
            std::streambuf* old = std::cout.rdbuf(ss.rdbuf());
            ArgsProbe probe(name == "multidiscrete");   // PPO_MultiDiscrete reads action_high / action_low, PPO_Discrete does not
            std::cout.rdbuf(old);
            std::vector<std::string> got;
            { std::string l; while (std::getline(ss, l)) got.push_back(l); }
            if (got != want_args) {
                std::fprintf(stderr, "getArgs output differs for variant %s:\n", name.c_str());
                for (auto& l : got) std::fprintf(stderr, "  got : %s\n", l.c_str());
                for (auto& l : want_args) std::fprintf(stderr, "  want: %s\n", l.c_str());
                return 1;
            }
            std::ostringstream fo;
            fo << std::setprecision(9);
            fo << "m_obs_size=" << probe.m_obs_size << "\nm_action_size=" << probe.m_action_size << "\nm_max_episode_steps=" << probe.m_max_episode_steps
               << "\nm_seed=" << probe.m_seed << "\nm_total_timesteps=" << probe.m_total_timesteps << "\nm_use_cuda=" << (probe.m_use_cuda ? 1 : 0)
               << "\nm_torch_deterministic=" << (probe.m_torch_deterministic ? 1 : 0) << "\nm_checkpoint_updates=" << probe.m_checkpoint_updates
               << "\nm_learning_rate=" << probe.m_learning_rate << "\nm_num_envs=" << probe.m_num_envs << "\nm_num_steps=" << probe.m_num_steps
               << "\nm_anneal_lr=" << (probe.m_anneal_lr ? 1 : 0) << "\nm_use_gae=" << (probe.m_use_gae ? 1 : 0) << "\nm_gamma=" << probe.m_gamma
               << "\nm_gae_lambda=" << probe.m_gae_lambda << "\nm_num_minibatches=" << probe.m_num_minibatches << "\nm_update_epochs=" << probe.m_update_epochs
               << "\nm_norm_adv=" << (probe.m_norm_adv ? 1 : 0) << "\nm_clip_coef=" << probe.m_clip_coef << "\nm_clip_vloss=" << (probe.m_clip_vloss ? 1 : 0)
               << "\nm_ent_coef=" << probe.m_ent_coef << "\nm_vf_coef=" << probe.m_vf_coef << "\nm_max_grad_norm=" << probe.m_max_grad_norm
               << "\nm_batch_size=" << probe.m_batch_size << "\nm_minibatch_size=" << probe.m_minibatch_size << "\n";
            std::string want_fields;
            for (auto& kv : fields) want_fields += kv.first + "=" + kv.second + "\n";
            if (fo.str() != want_fields) { std::fprintf(stderr, "fields differ for variant %s:\n%s--- want\n%s", name.c_str(), fo.str().c_str(), want_fields.c_str()); return 1; }
            if (name == "full") {   // a runnable configuration: the whole constructor, line by line
                std::stringstream cs;
                std::cout.copyfmt(std::ios(nullptr));
                std::streambuf* o2 = std::cout.rdbuf(cs.rdbuf());
                {
                    PPO_Discrete whole;
                }
                std::cout.rdbuf(o2);
                std::vector<std::string> g2, w2 = want_args;
                w2.insert(w2.end(), want_rest.begin(), want_rest.end());
                std::string l;
                while (std::getline(cs, l)) {
                    if (l.rfind("Warning: use_cuda", 0) == 0 || is_device_line(l)) continue;
                    g2.push_back(l);
                }
                if (g2 != w2) {
                    std::fprintf(stderr, "constructor output differs:\n");
                    for (auto& x : g2) std::fprintf(stderr, "  got : %s\n", x.c_str());
                    for (auto& x : w2) std::fprintf(stderr, "  want: %s\n", x.c_str());
                    return 1;
                }
            }
            variants++;
            return 0;
        };
        while (std::getline(gf, line)) {
            if (line.rfind("== variant ", 0) == 0) {
                if (check()) return 1;
                name = line.substr(11); section.clear(); toml.clear(); out.clear(); fields.clear();
            } else if (line == "-- toml" || line == "-- stdout" || line == "-- fields") {
                section = line.substr(3);
            } else if (section == "toml") toml += line + "\n";
            else if (section == "stdout") out += line + "\n";
            else if (section == "fields") { const size_t eq = line.find('='); fields.emplace_back(line.substr(0, eq), line.substr(eq + 1)); }
        }
        if (check()) return 1;
        REQUIRE(variants == 5);
        fs::current_path(scratch);
    }
    // ---- a short train() run with checkpoints against the reference's own run of the same PPOConfig.toml (tests/golden/train_run.txt): every
    //      console line outside the per-update table (the table carries wall-clock numbers) and the files left behind.  The thread-pool line
    //      names the machine's core count and the device line is this build's own.
    for (int which = 0; which < 2 && !golden.empty(); which++) {   // PPO_Discrete on CartPole, PPO_MultiDiscrete on MountainCar
        std::ifstream gf(golden + (which == 0 ? "/train_run.txt" : "/train_run_multidiscrete.txt"), std::ios::binary);
        REQUIRE(gf.good());
        std::map<std::string, std::string> sec;
        std::string line, cur;
        while (std::getline(gf, line)) { if (line.rfind("-- ", 0) == 0) { cur = line.substr(3); sec[cur]; } else sec[cur] += line + "\n"; }
        REQUIRE(sec.count("toml") && sec.count("phase1 constructor") && sec.count("phase1 train") && sec.count("phase1 files"));
        const fs::path dir = scratch / (which == 0 ? "train_run" : "train_run_md");
        fs::create_directories(dir);
        fs::current_path(dir);
        std::ofstream("PPOConfig.toml", std::ios::binary) << sec["toml"];
        auto filtered = [](const std::string& all) {
            std::istringstream is(all);
            std::string l, out;
            while (std::getline(is, l)) {
                if (l.empty() || l[0] == '-') continue;
                if (l[0] == '|') {   // table rows whose value does not depend on the wall clock or on the random stream
                    bool keep = false;
                    for (const char* k : { "iterations", "total_timesteps", "clip_range", "learning_rate", "n_updates" }) keep |= l.find(std::string("|    ") + k + " ") == 0;
                    if (!keep) continue;
                }
                if (l.rfind("Warning: use_cuda", 0) == 0 || l.rfind("Created ", 0) == 0) continue;
                if (l.rfind("Using ", 0) == 0 && l.find(" device") != std::string::npos) continue;
                out += l + "\n";
            }
            return out;
        };
        std::stringstream c1, t1;
        std::cout.copyfmt(std::ios(nullptr));
        std::streambuf* old = std::cout.rdbuf(c1.rdbuf());
        if (which == 0) {
            PPO_Discrete algo;
            std::cout.rdbuf(t1.rdbuf());
            algo.train();
        } else {
            PPO_MultiDiscrete algo;
            std::cout.rdbuf(t1.rdbuf());
            algo.train();
        }
        std::cout.rdbuf(old);
        const std::string gc = filtered(c1.str()), wc = filtered(sec["phase1 constructor"]), gt = filtered(t1.str()), wt = filtered(sec["phase1 train"]);
        if (gc != wc) { std::fprintf(stderr, "constructor lines differ:\n%s--- want\n%s", gc.c_str(), wc.c_str()); return 1; }
        if (gt != wt) { std::fprintf(stderr, "train() lines differ:\n%s--- want\n%s", gt.c_str(), wt.c_str()); return 1; }
        std::vector<std::string> names;
        for (const char* d : { "ModelCheckpoints", "OptimizerCheckpoints", "Models" })
            if (fs::exists(d)) for (auto const& e : fs::directory_iterator(d)) names.push_back(std::string(d) + "/" + e.path().filename().string());
        std::sort(names.begin(), names.end());
        std::string listing;
        for (auto& n : names) listing += n + "\n";
        if (listing != sec["phase1 files"]) { std::fprintf(stderr, "files differ:\n%s--- want\n%s", listing.c_str(), sec["phase1 files"].c_str()); return 1; }
        fs::current_path(scratch);
    }
    // ---- resume from files the REFERENCE wrote (tests/golden/ref_*_agent.pt / ref_*_optimizer.pt: what its train() left under ./Models/, copied
    //      out by oracle/ref_harness ptgold): the constructor picks them up from the checkpoint directories, the context then holds exactly the
    //      reference's parameters and AdamW state, and training goes on from the step count in the file name
    for (int which = 0; which < 2 && !golden.empty(); which++) {
        const std::string tag = which == 0 ? "discrete" : "multidiscrete";
        const fs::path dir = scratch / ("resume_ref_" + tag);
        fs::create_directories(dir / "ModelCheckpoints"); fs::create_directories(dir / "OptimizerCheckpoints");
        fs::current_path(dir);
        const int64_t steps_done = which == 0 ? 384 : 256;
        fs::copy_file(golden + "/ref_" + tag + "_agent.pt", "ModelCheckpoints/PPO_Agent_" + std::to_string(steps_done) + "_steps.pt");
        fs::copy_file(golden + "/ref_" + tag + "_optimizer.pt", "OptimizerCheckpoints/PPO_Optimizer_" + std::to_string(steps_done) + "_steps.pt");
        std::ofstream("PPOConfig.toml", std::ios::binary)
            << (which == 0 ? "[environment]\nobs_size = 4\naction_size = 2\nmax_episode_steps = 500\n[general]\nseed = 5\ntotal_timesteps = 640\ncheckpoint_updates = 100\n"
                             "[ppo]\nlearning_rate = 0.001\nnum_envs = 8\nnum_steps = 16\nanneal_lr = false\nnum_minibatches = 2\nupdate_epochs = 2\n"
                           : "[environment]\nobs_size = 2\naction_size = 3\naction_high = 1.0\naction_low = -1.0\nmax_episode_steps = 50\n[general]\nseed = 9\n"
                             "total_timesteps = 512\ncheckpoint_updates = 100\n[ppo]\nlearning_rate = 0.0005\nnum_envs = 4\nnum_steps = 32\nnum_minibatches = 4\nupdate_epochs = 1\n");
        const ppo::pt::AgentFile fa = ppo::pt::readAgent(golden + "/ref_" + tag + "_agent.pt");
        const ppo::pt::OptimizerFile fo = ppo::pt::readOptimizer(golden + "/ref_" + tag + "_optimizer.pt");
        std::vector<float> want_p, want_m, want_v;
        for (const auto& t : fa.tensors) want_p.insert(want_p.end(), t.values.begin(), t.values.end());
        for (const auto& t : fo.exp_avg) want_m.insert(want_m.end(), t.values.begin(), t.values.end());
        for (const auto& t : fo.exp_avg_sq) want_v.insert(want_v.end(), t.values.begin(), t.values.end());
        std::stringstream cap;
        std::streambuf* old = std::cout.rdbuf(cap.rdbuf());
        std::unique_ptr<PPOAlgorithm> algo;
        if (which == 0) algo = std::make_unique<PPO_Discrete>(); else algo = std::make_unique<PPO_MultiDiscrete>();
        std::cout.rdbuf(old);
        REQUIRE(cap.str().find("Loading model ./ModelCheckpoints/PPO_Agent_" + std::to_string(steps_done) + "_steps.pt...") != std::string::npos);
        REQUIRE(cap.str().find("Continuing training from step " + std::to_string(steps_done)) != std::string::npos);
        REQUIRE(cap.str().find("ignoring it") == std::string::npos);
        REQUIRE(algo->m_global_step == (uint64_t)steps_done);
        const int64_t P = ppo_param_count(algo->m_ctx);
        std::vector<float> p((size_t)P), m((size_t)P), v((size_t)P);
        int64_t step = 0;
        REQUIRE(ppo_params_get_h(algo->m_ctx, p.data(), P) == PPO_OK && ppo_optimizer_get_h(algo->m_ctx, m.data(), v.data(), P, &step) == PPO_OK);
        REQUIRE(want_p.size() == (size_t)P && std::memcmp(want_p.data(), p.data(), (size_t)P * 4) == 0);
        REQUIRE(std::memcmp(want_m.data(), m.data(), (size_t)P * 4) == 0 && std::memcmp(want_v.data(), v.data(), (size_t)P * 4) == 0);
        REQUIRE(step == fo.step[0] && step > 0);
        ppo_stats st{};
        REQUIRE(ppo_read_stats(algo->m_ctx, &st) == PPO_OK && st.learning_rate == fo.lr);   // torch::load(optimizer) brings the saved rate back (:834)
        old = std::cout.rdbuf(cap.rdbuf());
        algo->train();   // the remaining (total_timesteps - steps_done) / batch_size updates (:496)
        std::cout.rdbuf(old);
        const int64_t batch = which == 0 ? 128 : 128, updates = ((which == 0 ? 640 : 512) - steps_done) / batch;
        REQUIRE(algo->m_global_step == (uint64_t)(steps_done + updates * batch));
        REQUIRE(ppo_optimizer_get_h(algo->m_ctx, m.data(), v.data(), P, &step) == PPO_OK && step == fo.step[0] + updates * (which == 0 ? 4 : 4));
        REQUIRE(ppo_params_get_h(algo->m_ctx, p.data(), P) == PPO_OK);
        bool finite = true, moved = false;
        for (int64_t i = 0; i < P; i++) { finite = finite && std::isfinite(p[(size_t)i]); moved = moved || p[(size_t)i] != want_p[(size_t)i]; }
        REQUIRE(finite && moved);
        fs::current_path(scratch);
    }
    std::printf("HOST_FACADE_OK\n");
    return 0;
}
