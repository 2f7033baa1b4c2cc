// Entry point, same shape as the reference's driver.cpp:7-22: construct the algorithm, train, report exceptions.
//#include "PPO/PPO_MultiDiscrete.h"
#include "PPO/PPO_Discrete.h"

int main() {
    try {
        PPO_Discrete algo;
        algo.train();
    } catch (const std::exception& ex) {
        std::cerr << "Error Occured: " << ex.what() << std::endl;
        return 1;
    }
    return 0;
}
