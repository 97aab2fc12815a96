// input_group.h -- reader of the reference's input-file grammar (ITensor InputGroup as used by
// fixedL.cc:584-608; sample: sample_inputs/input_fixedL).
//
//   input
//   {
//   key = value      (any number per line, any indentation; unknown keys are ignored)
//   }
//
// getX(key, default) looks for the exact, case-sensitive key followed by '=' inside the named group and
// parses the next whitespace-delimited token; a missing key yields the default.
#pragma once
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace tnmlh {

class InputGroup {
  public:
    InputGroup(const std::string& file, const std::string& group) : file_(file), group_(group) {
        std::ifstream in(file);
        if (!in) throw std::runtime_error("Couldn't open input file " + file);
        std::stringstream ss; ss << in.rdbuf();
        const std::string txt = ss.str();
        // locate the token `group` followed by '{'
        size_t pos = 0;
        bool found = false;
        while ((pos = txt.find(group, pos)) != std::string::npos) {
            const bool left_ok = pos == 0 || isspace((unsigned char)txt[pos - 1]);
            size_t q = pos + group.size();
            while (q < txt.size() && isspace((unsigned char)txt[q])) ++q;
            if (left_ok && q < txt.size() && txt[q] == '{') { pos = q + 1; found = true; break; }
            pos += group.size();
        }
        if (!found) throw std::runtime_error("Couldn't find group '" + group + "' in " + file);
        const size_t end = txt.find('}', pos);
        if (end == std::string::npos) throw std::runtime_error("Unterminated group '" + group + "' in " + file);
        std::string body = txt.substr(pos, end - pos);
        // make '=' its own token
        std::string spaced;
        for (char ch : body) { if (ch == '=') spaced += " = "; else spaced += ch; }
        std::istringstream ts(spaced);
        std::vector<std::string> tok;
        for (std::string t; ts >> t;) tok.push_back(t);
        for (size_t i = 0; i + 2 < tok.size();) {
            if (tok[i + 1] == "=" && tok[i] != "=" && tok[i + 2] != "=") { kv_[tok[i]] = tok[i + 2]; i += 3; }
            else ++i;
        }
    }
    bool has(const std::string& key) const { return kv_.count(key) != 0; }
    std::string getString(const std::string& key, const std::string& def) const { auto it = kv_.find(key); return it == kv_.end() ? def : it->second; }
    long getInt(const std::string& key, long def) const {
        auto it = kv_.find(key); if (it == kv_.end()) return def;
        try { return std::stol(it->second); } catch (...) { throw std::runtime_error("Input key " + key + ": expected an integer, got '" + it->second + "'"); }
    }
    double getReal(const std::string& key, double def) const {
        auto it = kv_.find(key); if (it == kv_.end()) return def;
        try { return std::stod(it->second); } catch (...) { throw std::runtime_error("Input key " + key + ": expected a real, got '" + it->second + "'"); }
    }
    bool getYesNo(const std::string& key, bool def) const {           // yes/no by first letter, like ITensor
        auto it = kv_.find(key); if (it == kv_.end() || it->second.empty()) return def;
        const char c0 = it->second[0];
        if (c0 == 'y' || c0 == 'Y' || c0 == 't' || c0 == 'T' || c0 == '1') return true;
        if (c0 == 'n' || c0 == 'N' || c0 == 'f' || c0 == 'F' || c0 == '0') return false;
        throw std::runtime_error("Input key " + key + ": expected yes/no, got '" + it->second + "'");
    }
    const std::map<std::string, std::string>& all() const { return kv_; }

  private:
    std::string file_, group_;
    std::map<std::string, std::string> kv_;
};

}  // namespace tnmlh
