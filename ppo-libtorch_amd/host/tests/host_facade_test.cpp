// Exercises the C++ classes with the reference's names on a real GPU (run by tests/test_host_facade.py, -m gpu).
// Prints "HOST_FACADE_OK" when every check holds.
#include <cmath>
#include <cstdio>
#include <filesystem>
#include <fstream>

#include "../PPO/PPO_Discrete.h"
#include "../PPO/PPO_MultiDiscrete.h"

#define REQUIRE(cond)                                                                 \
    do {                                                                              \
        if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
    } while (0)

static void writeConfig(const char* body) { std::ofstream("PPOConfig.toml") << body; }

int main() {
    namespace fs = std::filesystem;
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
    }
    {
        PPO_Discrete resumed;   // picks up the newest checkpoint (mtime) and its step count from the file name
        REQUIRE(resumed.m_global_step == 32768);
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
    std::printf("HOST_FACADE_OK\n");
    return 0;
}
