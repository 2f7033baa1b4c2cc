// Reader and writer for the checkpoint files the reference leaves behind: `torch::save(m_agent, ...)` and `torch::save(*m_optimizer, ...)`
// (PPO/PPO_Discrete.cpp:662-673, 679-686; read back by loadPolicyFromCheckpoint, :782-835).  Those are LibTorch "script module" archives: a
// ZIP file with stored (uncompressed) records
//     <stem>/data.pkl      pickle (protocol 2) of the module object: nested `__torch__...Module` objects whose state dictionaries hold
//                          sub-modules, tensors (`torch._utils._rebuild_tensor_v2` over a persistent storage id), ints, floats, strings
//     <stem>/data/<key>    the raw little-endian bytes of storage <key>
//     <stem>/code/...      TorchScript class declarations of every module object (attribute names and types), constants.pkl, version
// This file restates that container from its on-disk form (fixtures written by the compiled reference: tests/golden/ref_*_agent.pt,
// ref_*_optimizer.pt) in plain C++17: no LibTorch, no zlib.  What is read: every record that is stored (method 0) with its CRC checked, and
// the pickle opcodes LibTorch's pickler emits.  What is written: the same layout, accepted by `torch::load` / `torch.jit.load`.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace ppo {
namespace pt {

struct NamedTensor {
    std::string name;               // dotted path from the root module, e.g. "m_Critic.criticInputLayer.weight"
    std::vector<int64_t> sizes;
    std::vector<float> values;      // row-major, contiguous
};

// Agent file: the parameters in the order the module tree stores them (= Agent::parameters() order: critic first, Agent.cpp:65-66)
struct AgentFile {
    std::vector<NamedTensor> tensors;
};

// Optimizer file (torch::optim::AdamW, "pytorch_version" 1.5.0 layout): state per parameter in param_groups[0].params order
struct OptimizerFile {
    std::vector<int64_t> step;                  // one per parameter
    std::vector<NamedTensor> exp_avg, exp_avg_sq;
    double lr = 0.0, beta1 = 0.9, beta2 = 0.999, eps = 1e-8, weight_decay = 0.01;
    bool amsgrad = false;
};

// true when the file starts like a ZIP archive (what distinguishes the reference's checkpoints from this build's own flat format)
bool isTorchArchive(const std::string& path);

// Both throw std::runtime_error naming the file and what is wrong with it (not a ZIP, compressed or damaged record, unknown pickle opcode,
// a layout other than the one above).
AgentFile readAgent(const std::string& path);
OptimizerFile readOptimizer(const std::string& path);

// The reference's Agent (Agent.cpp:25-66): m_Critic { criticInputLayer, Tanh1, criticMiddleLayer, Tanh2, criticOutputLayer } then m_Actor
// { actorInputLayer, Tanh1, actorMiddleLayer, Tanh2, actorOutputLayer }; `flat` holds weight, bias of the six Linear layers in that order.
// `stem` names the archive's internal directory (LibTorch uses the file's name without extension; empty: derived from `path`).
void writeAgent(const std::string& path, int64_t obs, int64_t hidden, int64_t act, const std::vector<float>& flat, const std::string& stem = "");
// AdamW state for those twelve parameters, every one at the same step count
void writeOptimizer(const std::string& path, int64_t obs, int64_t hidden, int64_t act, const std::vector<float>& exp_avg,
                    const std::vector<float>& exp_avg_sq, int64_t step, double lr, double eps, double weight_decay, const std::string& stem = "");

// element counts of the twelve parameters in file order: {hidden*obs, hidden, hidden*hidden, hidden, 1*hidden, 1, hidden*obs, hidden, hidden*hidden, hidden, act*hidden, act}
std::vector<std::vector<int64_t>> agentShapes(int64_t obs, int64_t hidden, int64_t act);

}  // namespace pt
}  // namespace ppo
