// Command-line face of Utils/TorchArchive for tests/test_torch_archive.py (needs neither a GPU nor libppo_hip.so):
//   torch_archive_tool dump-agent <file.pt>        name, sizes, then every value as the hex of its float bits
//   torch_archive_tool dump-optimizer <file.pt>    options, then per parameter: step, exp_avg, exp_avg_sq
//   torch_archive_tool rewrite <agent_in> <optimizer_in> <agent_out> <optimizer_out> <obs> <act>
//                                                  reads the two archives and writes them again with this build's writer
#include <cinttypes>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <stdexcept>

#include "Utils/TorchArchive.h"

namespace pt = ppo::pt;

static void dumpTensor(const char* tag, const pt::NamedTensor& t) {
    std::printf("%s %s [", tag, t.name.c_str());
    for (size_t i = 0; i < t.sizes.size(); i++) std::printf(i ? ",%" PRId64 : "%" PRId64, t.sizes[i]);
    std::printf("]");
    for (float v : t.values) { uint32_t u; std::memcpy(&u, &v, 4); std::printf(" %08x", u); }
    std::printf("\n");
}

int main(int argc, char** argv) {
    try {
        const std::string mode = argc > 1 ? argv[1] : "";
        if (mode == "dump-agent" && argc == 3) {
            for (const auto& t : pt::readAgent(argv[2]).tensors) dumpTensor("tensor", t);
            return 0;
        }
        if (mode == "dump-optimizer" && argc == 3) {
            const pt::OptimizerFile f = pt::readOptimizer(argv[2]);
            std::printf("options lr=%a beta1=%a beta2=%a eps=%a weight_decay=%a amsgrad=%d\n", f.lr, f.beta1, f.beta2, f.eps, f.weight_decay, f.amsgrad ? 1 : 0);
            for (size_t i = 0; i < f.step.size(); i++) {
                std::printf("step %zu %" PRId64 "\n", i, f.step[i]);
                dumpTensor("exp_avg", f.exp_avg[i]);
                dumpTensor("exp_avg_sq", f.exp_avg_sq[i]);
            }
            return 0;
        }
        if (mode == "rewrite" && argc == 8) {
            const int64_t obs = std::atoll(argv[6]), act = std::atoll(argv[7]);
            const pt::AgentFile a = pt::readAgent(argv[2]);
            const pt::OptimizerFile o = pt::readOptimizer(argv[3]);
            std::vector<float> p, m, v;
            for (const auto& t : a.tensors) p.insert(p.end(), t.values.begin(), t.values.end());
            for (const auto& t : o.exp_avg) m.insert(m.end(), t.values.begin(), t.values.end());
            for (const auto& t : o.exp_avg_sq) v.insert(v.end(), t.values.begin(), t.values.end());
            pt::writeAgent(argv[4], obs, 64, act, p);
            pt::writeOptimizer(argv[5], obs, 64, act, m, v, o.step.at(0), o.lr, o.eps, o.weight_decay);
            return 0;
        }
        std::cerr << "usage: torch_archive_tool dump-agent <pt> | dump-optimizer <pt> | rewrite <agent_in> <opt_in> <agent_out> <opt_out> <obs> <act>\n";
        return 2;
    } catch (const std::exception& ex) {
        std::cerr << ex.what() << std::endl;
        return 1;
    }
}
