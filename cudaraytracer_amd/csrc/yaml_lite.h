// Minimal YAML reader for scene files: block maps and sequences, flow sequences/maps,
// plain / quoted scalars, comments.  It covers what the reference's scene loader reads
// through yaml-cpp (src/scene.cuh:152-690, src/utility.cuh:294-325) and keeps key order.
#ifndef CRT_YAML_LITE_H
#define CRT_YAML_LITE_H

#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace yamllite {

struct ParseError : std::runtime_error {
    int line;
    ParseError(const std::string& what, int line_) : std::runtime_error(what), line(line_) {}
};

struct Node {
    enum Kind { Null, Scalar, Seq, Map };
    Kind kind = Null;
    int line = 0;                      // 0-based source line, like YAML::Mark().line
    std::string scalar;
    bool quoted = false;
    std::vector<Node> seq;
    std::vector<std::pair<std::string, Node>> map;

    bool defined() const { return kind != Null; }
    bool is_scalar() const { return kind == Scalar; }
    bool is_seq() const { return kind == Seq; }
    bool is_map() const { return kind == Map; }
    // Returns a Null node when the key is absent (yaml-cpp's "zombie" node).
    const Node& operator[](const std::string& key) const;
    const Node& operator[](size_t i) const;
    size_t size() const { return kind == Seq ? seq.size() : (kind == Map ? map.size() : 0); }

    // Conversions throw ParseError when the scalar does not convert (YAML::BadConversion).
    std::string as_string() const;
    float as_float() const;
    int as_int() const;
    bool as_bool() const;
};

Node parse(const std::string& text);
Node parse_file(const std::string& path);

}  // namespace yamllite

#endif
