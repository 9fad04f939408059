// Minimal stand-in for the subset of yaml-cpp the reference's drivers use (YAML::LoadFile, Node::operator[],
// Node::as<T>, IsDefined): yaml-cpp is not installed in this image.  Parses the
// indentation-based maps of config/nice_slam.yaml and config/cofusion.yaml (scalars, nested maps, comments).
// If the real yaml-cpp is available, drop this directory from the include path -- the host classes only use
// the calls listed above.
#pragma once
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace YAML {

class Node {
  public:
    Node() : d_(std::make_shared<Data>()) {}
    bool IsDefined() const { return d_->defined; }
    bool IsMap() const { return !d_->kids.empty(); }
    Node operator[](const std::string& key) const
    {
        auto it = d_->kids.find(key);
        if (it == d_->kids.end()) return Node::undefined(key);
        return it->second;
    }
    template <typename T> T as() const
    {
        if (!d_->defined) throw std::runtime_error("yaml: key '" + d_->scalar + "' is not defined");
        return convert<T>(d_->scalar);
    }
    // builder access (used by the parser and by tests that assemble configs in memory)
    Node& set(const std::string& key, const Node& n) { d_->kids[key] = n; d_->defined = true; return *this; }
    static Node scalar(const std::string& v) { Node n; n.d_->defined = true; n.d_->scalar = v; return n; }

  private:
    struct Data { bool defined = false; std::string scalar; std::map<std::string, Node> kids; };
    std::shared_ptr<Data> d_;
    static Node undefined(const std::string& key) { Node n; n.d_->scalar = key; return n; }
    template <typename T> static T convert(const std::string& s);
};

template <> inline std::string Node::convert<std::string>(const std::string& s) { return s; }
template <> inline int Node::convert<int>(const std::string& s) { return (int)std::stod(s); }
template <> inline float Node::convert<float>(const std::string& s) { return (float)std::stod(s); }
template <> inline double Node::convert<double>(const std::string& s) { return std::stod(s); }
template <> inline bool Node::convert<bool>(const std::string& s)
{
    return s == "True" || s == "true" || s == "TRUE" || s == "yes" || s == "1" || s == "on";
}

inline Node Load(std::istream& in)
{
    struct Frame { int indent; Node node; };
    Node root;
    std::vector<Frame> stack{{-1, root}};
    std::string line;
    while (std::getline(in, line)) {
        bool in_s = false, in_d = false;                      // strip comments outside quotes
        for (size_t i = 0; i < line.size(); ++i) {
            if (line[i] == '\'' && !in_d) in_s = !in_s;
            else if (line[i] == '"' && !in_s) in_d = !in_d;
            else if (line[i] == '#' && !in_s && !in_d) { line.erase(i); break; }
        }
        size_t a = line.find_first_not_of(" \t");
        if (a == std::string::npos) continue;
        size_t b = line.find_last_not_of(" \t\r\n");
        std::string body = line.substr(a, b - a + 1);
        size_t colon = body.find(':');
        if (colon == std::string::npos) continue;
        std::string key = body.substr(0, colon), val = body.substr(colon + 1);
        size_t v0 = val.find_first_not_of(" \t");
        val = v0 == std::string::npos ? "" : val.substr(v0);
        if (val.size() >= 2 && ((val.front() == '\'' && val.back() == '\'') || (val.front() == '"' && val.back() == '"')))
            val = val.substr(1, val.size() - 2);
        int indent = (int)a;
        while (stack.size() > 1 && stack.back().indent >= indent) stack.pop_back();
        if (val.empty()) {
            Node child;
            stack.back().node.set(key, child);
            stack.push_back({indent, child});
        } else {
            stack.back().node.set(key, Node::scalar(val));
        }
    }
    return root;
}

inline Node LoadFile(const std::string& path)
{
    std::ifstream f(path);
    if (!f) throw std::runtime_error("yaml: cannot open " + path);
    return Load(f);
}

}  // namespace YAML
