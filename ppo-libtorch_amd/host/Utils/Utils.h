// Drop-in for the reference's Utils/Utils.h: PPOUtils statics (:13-21) and the 100-episode CircularBuffer (:30-79).
#pragma once
#include <algorithm>
#include <cctype>
#include <cstdint>
#include <string>
#include <vector>

class PPOUtils {
  public:
    static float getVectorMean(std::vector<float> v) {          // Utils.cpp:5-14 (float running total)
        float total = 0.f;
        for (float x : v) total += x;
        return total / v.size();
    }
    static std::string formatString(std::string& str) {          // strip blanks, lower-case (Utils.cpp:17-27)
        str.erase(std::remove(str.begin(), str.end(), ' '), str.end());
        for (char& ch : str) ch = static_cast<char>(std::tolower(static_cast<unsigned char>(ch)));
        return str;
    }
    // digits that follow fileHead in a checkpoint file name (Utils.cpp:38-60): "PPO_Agent_4096_steps.pt" -> "4096"
    static std::string getLoadFromSteps(std::string const& str, std::string const& fileHead) {
        const std::size_t at = str.find(fileHead);
        if (at == std::string::npos) return {};
        std::size_t b = at + fileHead.size(), e = b;
        while (e < str.size() && std::isdigit(static_cast<unsigned char>(str[e]))) e++;
        return str.substr(b, e - b);
    }
    static bool isNumber(const std::string& s) {
        return !s.empty() && std::all_of(s.begin(), s.end(), [](unsigned char ch) { return std::isdigit(ch) != 0; });
    }
};

// Ring of the last `capacity` finished episodes with running sums (SB3 keeps 100 regardless of the env count).
class CircularBuffer {
  public:
    explicit CircularBuffer(size_t capacity) : m_rew(capacity), m_len(capacity), m_cap(capacity) {}
    void add(float reward, int64_t length) {
        if (m_n == m_cap) { m_rsum -= m_rew[m_at]; m_lsum -= static_cast<double>(m_len[m_at]); } else { m_n++; }
        m_rew[m_at] = reward;
        m_len[m_at] = length;
        m_rsum += reward;
        m_lsum += static_cast<double>(length);
        m_at = (m_at + 1) % m_cap;
    }
    bool empty() const { return m_n == 0; }
    size_t size() const { return m_n; }
    float avgReward() const { return m_n ? static_cast<float>(m_rsum / m_n) : 0.0f; }
    double avgLength() const { return m_n ? m_lsum / m_n : 0.0; }
    // device-side ring (PPO_BUF_FIN_* -> ppo_read_stats) already holds the averages: mirror them
    void assign(double avg_len, float avg_rew, size_t count) { m_n = count; m_lsum = avg_len * count; m_rsum = static_cast<double>(avg_rew) * count; m_mirror = true; }

  private:
    std::vector<float> m_rew;
    std::vector<int64_t> m_len;
    size_t m_cap, m_n = 0, m_at = 0;
    double m_rsum = 0.0, m_lsum = 0.0;
    bool m_mirror = false;
};
