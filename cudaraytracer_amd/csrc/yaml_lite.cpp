#include "yaml_lite.h"

#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <sstream>

namespace yamllite {

namespace {

const Node k_null_node;

struct Line {
    int indent;
    std::string text;   // content without indentation, comments and trailing blanks
    int number;         // 0-based
};

// Strip a trailing comment: '#' at line start or preceded by whitespace, outside quotes.
std::string strip_comment(const std::string& s) {
    char quote = 0;
    for (size_t i = 0; i < s.size(); i++) {
        char c = s[i];
        if (quote) {
            if (c == quote) {
                if (quote == '\'' && i + 1 < s.size() && s[i + 1] == '\'') { i++; continue; }
                quote = 0;
            } else if (quote == '"' && c == '\\') {
                i++;
            }
        } else if (c == '"' || c == '\'') {
            // a quote only opens a quoted scalar at the start of a token
            if (i == 0 || s[i - 1] == ' ' || s[i - 1] == '[' || s[i - 1] == ',' || s[i - 1] == '{' || s[i - 1] == ':') quote = c;
        } else if (c == '#' && (i == 0 || s[i - 1] == ' ' || s[i - 1] == '\t')) {
            return s.substr(0, i);
        }
    }
    return s;
}

std::string rtrim(std::string s) {
    while (!s.empty() && (s.back() == ' ' || s.back() == '\t' || s.back() == '\r' || s.back() == '\n')) s.pop_back();
    return s;
}
std::string trim(const std::string& s) {
    size_t a = 0;
    while (a < s.size() && (s[a] == ' ' || s[a] == '\t')) a++;
    return rtrim(s.substr(a));
}

// Net bracket depth of a line outside quotes (to join multi-line flow collections).
int bracket_balance(const std::string& s) {
    int depth = 0;
    char quote = 0;
    for (size_t i = 0; i < s.size(); i++) {
        char c = s[i];
        if (quote) {
            if (c == quote) quote = 0;
            else if (quote == '"' && c == '\\') i++;
        } else if (c == '"' || c == '\'') quote = c;
        else if (c == '[' || c == '{') depth++;
        else if (c == ']' || c == '}') depth--;
    }
    return depth;
}

std::vector<Line> split_lines(const std::string& text) {
    std::vector<Line> out;
    std::istringstream in(text);
    std::string raw;
    int number = 0;
    while (std::getline(in, raw)) {
        int this_line = number++;
        std::string s = rtrim(strip_comment(raw));
        size_t ind = 0;
        while (ind < s.size() && s[ind] == ' ') ind++;
        if (ind < s.size() && s[ind] == '\t') throw ParseError("tabs are not allowed for indentation", this_line);
        if (ind == s.size()) continue;   // blank
        std::string body = s.substr(ind);
        if (body == "---" && out.empty()) continue;   // document start marker
        if (body == "...") break;
        // join continuation lines of an unbalanced flow collection
        int bal = bracket_balance(body);
        while (bal > 0 && std::getline(in, raw)) {
            number++;
            std::string more = trim(strip_comment(raw));
            body += " " + more;
            bal = bracket_balance(body);
        }
        out.push_back({(int)ind, body, this_line});
    }
    return out;
}

Node make_scalar(const std::string& raw, int line) {
    Node n;
    n.line = line;
    std::string s = trim(raw);
    if (s.empty() || s == "~" || s == "null" || s == "Null" || s == "NULL") return n;   // Null
    n.kind = Node::Scalar;
    if (s.size() >= 2 && s.front() == '"' && s.back() == '"') {
        n.quoted = true;
        std::string out;
        for (size_t i = 1; i + 1 < s.size(); i++) {
            if (s[i] == '\\' && i + 2 < s.size()) {
                char c = s[++i];
                switch (c) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case '\\': out += '\\'; break;
                case '"': out += '"'; break;
                case '/': out += '/'; break;
                default: out += '\\'; out += c;
                }
            } else out += s[i];
        }
        n.scalar = out;
    } else if (s.size() >= 2 && s.front() == '\'' && s.back() == '\'') {
        n.quoted = true;
        std::string out;
        for (size_t i = 1; i + 1 < s.size(); i++) {
            if (s[i] == '\'' && s[i + 1] == '\'' && i + 2 < s.size()) { out += '\''; i++; }
            else out += s[i];
        }
        n.scalar = out;
    } else {
        n.scalar = s;
    }
    return n;
}

// Split the inside of a flow collection at top-level commas.
std::vector<std::string> split_flow(const std::string& s, int line) {
    std::vector<std::string> parts;
    std::string cur;
    int depth = 0;
    char quote = 0;
    for (size_t i = 0; i < s.size(); i++) {
        char c = s[i];
        if (quote) {
            cur += c;
            if (c == quote) quote = 0;
            else if (quote == '"' && c == '\\' && i + 1 < s.size()) cur += s[++i];
            continue;
        }
        if (c == '"' || c == '\'') { quote = c; cur += c; continue; }
        if (c == '[' || c == '{') depth++;
        if (c == ']' || c == '}') depth--;
        if (c == ',' && depth == 0) { parts.push_back(cur); cur.clear(); continue; }
        cur += c;
    }
    if (depth != 0 || quote) throw ParseError("unterminated flow collection", line);
    if (!trim(cur).empty() || !parts.empty()) parts.push_back(cur);
    return parts;
}

// Position of the ": " (or trailing ':') that separates key and value, or npos.
size_t find_map_colon(const std::string& s) {
    char quote = 0;
    int depth = 0;
    for (size_t i = 0; i < s.size(); i++) {
        char c = s[i];
        if (quote) {
            if (c == quote) quote = 0;
            else if (quote == '"' && c == '\\') i++;
            continue;
        }
        if ((c == '"' || c == '\'') && (i == 0 || s[i - 1] == ' ')) { quote = c; continue; }
        if (c == '[' || c == '{') depth++;
        else if (c == ']' || c == '}') depth--;
        else if (c == ':' && depth == 0 && (i + 1 == s.size() || s[i + 1] == ' ')) return i;
    }
    return std::string::npos;
}

Node parse_inline(const std::string& raw, int line);

Node parse_flow_seq(const std::string& s, int line) {
    Node n;
    n.kind = Node::Seq;
    n.line = line;
    for (const std::string& part : split_flow(s.substr(1, s.size() - 2), line)) n.seq.push_back(parse_inline(part, line));
    return n;
}
Node parse_flow_map(const std::string& s, int line) {
    Node n;
    n.kind = Node::Map;
    n.line = line;
    for (const std::string& part : split_flow(s.substr(1, s.size() - 2), line)) {
        std::string p = trim(part);
        if (p.empty()) continue;
        size_t c = find_map_colon(p);
        if (c == std::string::npos) throw ParseError("expected 'key: value' in flow map", line);
        Node key = make_scalar(p.substr(0, c), line);
        n.map.emplace_back(key.scalar, parse_inline(p.substr(c + 1), line));
    }
    return n;
}
Node parse_inline(const std::string& raw, int line) {
    std::string s = trim(raw);
    if (s.size() >= 2 && s.front() == '[' && s.back() == ']') return parse_flow_seq(s, line);
    if (s.size() >= 2 && s.front() == '{' && s.back() == '}') return parse_flow_map(s, line);
    return make_scalar(s, line);
}

bool is_seq_item(const std::string& t) { return t == "-" || (t.size() >= 2 && t[0] == '-' && t[1] == ' '); }

struct Parser {
    std::vector<Line> lines;
    size_t pos = 0;

    Node parse_block(int indent) {
        if (pos >= lines.size()) return Node();
        if (is_seq_item(lines[pos].text)) return parse_seq(indent);
        return parse_map(indent);
    }

    Node parse_seq(int indent) {
        Node n;
        n.kind = Node::Seq;
        n.line = lines[pos].number;
        while (pos < lines.size() && lines[pos].indent == indent && is_seq_item(lines[pos].text)) {
            Line& ln = lines[pos];
            std::string rest = ln.text.size() > 1 ? ln.text.substr(1) : std::string();
            size_t lead = 0;
            while (lead < rest.size() && rest[lead] == ' ') lead++;
            rest = rest.substr(lead);
            if (rest.empty()) {
                pos++;
                if (pos < lines.size() && lines[pos].indent > indent) n.seq.push_back(parse_block(lines[pos].indent));
                else n.seq.push_back(Node());
                continue;
            }
            bool flow = rest[0] == '[' || rest[0] == '{';
            if (!flow && (find_map_colon(rest) != std::string::npos || is_seq_item(rest))) {
                // the item is a block collection whose first line shares the dash's line
                ln.indent = indent + 1 + (int)lead;
                ln.text = rest;
                n.seq.push_back(parse_block(ln.indent));
            } else {
                n.seq.push_back(parse_inline(rest, ln.number));
                pos++;
            }
        }
        if (pos < lines.size() && lines[pos].indent > indent) throw ParseError("unexpected indentation", lines[pos].number);
        return n;
    }

    Node parse_map(int indent) {
        Node n;
        n.kind = Node::Map;
        n.line = lines[pos].number;
        while (pos < lines.size() && lines[pos].indent == indent && !is_seq_item(lines[pos].text)) {
            const Line ln = lines[pos];
            size_t c = find_map_colon(ln.text);
            if (c == std::string::npos) throw ParseError("expected 'key: value'", ln.number);
            Node key = make_scalar(ln.text.substr(0, c), ln.number);
            std::string rest = trim(ln.text.substr(c + 1));
            pos++;
            Node value;
            if (!rest.empty()) {
                value = parse_inline(rest, ln.number);
            } else if (pos < lines.size() && lines[pos].indent > indent) {
                value = parse_block(lines[pos].indent);
            } else if (pos < lines.size() && lines[pos].indent == indent && is_seq_item(lines[pos].text)) {
                value = parse_seq(indent);   // "key:" followed by a sequence at the same indentation
            } else {
                value.line = ln.number;
            }
            n.map.emplace_back(key.scalar, std::move(value));
        }
        if (pos < lines.size() && lines[pos].indent > indent) throw ParseError("unexpected indentation", lines[pos].number);
        return n;
    }
};

}  // namespace

const Node& Node::operator[](const std::string& key) const {
    if (kind == Map)
        for (const auto& kv : map)
            if (kv.first == key) return kv.second;
    return k_null_node;
}
const Node& Node::operator[](size_t i) const {
    if (kind == Seq && i < seq.size()) return seq[i];
    return k_null_node;
}

std::string Node::as_string() const {
    if (kind != Scalar) throw ParseError("bad conversion: not a scalar", line);
    return scalar;
}
float Node::as_float() const {
    if (kind != Scalar) throw ParseError("bad conversion: not a scalar", line);
    const std::string& s = scalar;
    if (s == ".inf" || s == ".Inf" || s == ".INF" || s == "+.inf") return std::numeric_limits<float>::infinity();
    if (s == "-.inf" || s == "-.Inf" || s == "-.INF") return -std::numeric_limits<float>::infinity();
    if (s == ".nan" || s == ".NaN" || s == ".NAN") return std::numeric_limits<float>::quiet_NaN();
    if (s.empty()) throw ParseError("bad conversion: empty scalar", line);
    char* end = nullptr;
    float v = std::strtof(s.c_str(), &end);
    if (end == s.c_str() || *end != '\0') throw ParseError("bad conversion: '" + s + "' is not a number", line);
    return v;
}
int Node::as_int() const {
    if (kind != Scalar) throw ParseError("bad conversion: not a scalar", line);
    const std::string& s = scalar;
    if (s.empty()) throw ParseError("bad conversion: empty scalar", line);
    char* end = nullptr;
    errno = 0;
    long v = std::strtol(s.c_str(), &end, 0);
    if (end == s.c_str() || *end != '\0' || errno == ERANGE) throw ParseError("bad conversion: '" + s + "' is not an integer", line);
    return (int)v;
}
bool Node::as_bool() const {
    if (kind != Scalar) throw ParseError("bad conversion: not a scalar", line);
    std::string s;
    for (char c : scalar) s += (char)std::tolower((unsigned char)c);
    if (s == "true" || s == "yes" || s == "y" || s == "on") return true;
    if (s == "false" || s == "no" || s == "n" || s == "off") return false;
    throw ParseError("bad conversion: '" + scalar + "' is not a boolean", line);
}

Node parse(const std::string& text) {
    Parser p;
    p.lines = split_lines(text);
    if (p.lines.empty()) return Node();
    Node root = p.parse_block(p.lines[0].indent);
    if (p.pos < p.lines.size()) throw ParseError("unexpected content", p.lines[p.pos].number);
    return root;
}

Node parse_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw ParseError("bad file: " + path, 0);
    std::ostringstream ss;
    ss << f.rdbuf();
    return parse(ss.str());
}

}  // namespace yamllite
