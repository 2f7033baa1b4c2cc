// Reader for the flat subset of TOML the reference's ./PPOConfig.toml uses (three [sections] of `key = scalar` lines with
// '#' comments; reference PPO_Discrete.cpp:107-255 reads it through toml++).  Keys and sections are the reference's.
#pragma once
#include <algorithm>
#include <fstream>
#include <map>
#include <optional>
#include <sstream>
#include <stdexcept>
#include <string>

namespace ppo {

class FlatToml {
  public:
    struct ParseError : std::runtime_error {
        int line;
        ParseError(const std::string& what, int ln) : std::runtime_error(what), line(ln) {}
    };
    static bool exists(const std::string& path) { return std::ifstream(path).good(); }
    explicit FlatToml(const std::string& path) {
        std::ifstream f(path);
        if (!f) throw std::runtime_error("cannot open " + path);
        std::string line, section;
        int ln = 0;
        while (std::getline(f, line)) {
            ln++;
            const auto hash = line.find('#');
            if (hash != std::string::npos) line.erase(hash);
            trim(line);
            if (line.empty()) continue;
            if (line.front() == '[') {
                if (line.back() != ']') throw ParseError("unterminated section header", ln);
                section = line.substr(1, line.size() - 2);
                trim(section);
                continue;
            }
            const auto eq = line.find('=');
            if (eq == std::string::npos) throw ParseError("expected key = value", ln);
            std::string key = line.substr(0, eq), val = line.substr(eq + 1);
            trim(key); trim(val);
            if (key.empty() || val.empty()) throw ParseError("empty key or value", ln);
            m_values[section + "." + key] = val;
        }
    }
    std::optional<int64_t> integer(const std::string& section, const std::string& key) const {
        auto v = raw(section, key);
        if (!v) return std::nullopt;
        try { size_t n = 0; std::string s = strip_underscores(*v); int64_t x = std::stoll(s, &n); if (n == s.size()) return x; } catch (...) {}
        return std::nullopt;
    }
    std::optional<float> real(const std::string& section, const std::string& key) const {  // the reference reads floats with value<float>()
        auto v = raw(section, key);
        if (!v) return std::nullopt;
        try { size_t n = 0; std::string s = strip_underscores(*v); double x = std::stod(s, &n); if (n == s.size()) return static_cast<float>(x); } catch (...) {}
        return std::nullopt;
    }
    std::optional<bool> boolean(const std::string& section, const std::string& key) const {
        auto v = raw(section, key);
        if (!v) return std::nullopt;
        if (*v == "true") return true;
        if (*v == "false") return false;
        return std::nullopt;
    }

  private:
    std::optional<std::string> raw(const std::string& section, const std::string& key) const {
        auto it = m_values.find(section + "." + key);
        if (it == m_values.end()) return std::nullopt;
        return it->second;
    }
    static std::string strip_underscores(std::string s) { s.erase(std::remove(s.begin(), s.end(), '_'), s.end()); return s; }
    static void trim(std::string& s) {
        const char* ws = " \t\r\n";
        s.erase(0, s.find_first_not_of(ws));
        const auto e = s.find_last_not_of(ws);
        if (e != std::string::npos) s.erase(e + 1); else s.clear();
    }
    std::map<std::string, std::string> m_values;
};

}  // namespace ppo
