// json.hpp -- a small JSON DOM parser for the scene loader (serde_json's role in parser.rs:249).
// Objects keep insertion order; duplicate keys keep the LAST value (serde_json's behaviour for maps).
#pragma once
#include <cctype>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace mi355rt_host {

struct JsonValue {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0.0;
    bool integral = false;      // Number: the token had no fraction and no exponent and no minus sign (what serde accepts for usize)
    std::string str;
    std::vector<JsonValue> arr;
    std::vector<std::pair<std::string, JsonValue>> obj;

    bool is_null() const { return kind == Null; }
    bool is_number() const { return kind == Number; }
    bool is_string() const { return kind == String; }
    bool is_array() const { return kind == Array; }
    bool is_object() const { return kind == Object; }
    const JsonValue* get(const std::string& key) const {
        if (kind != Object) return nullptr;
        const JsonValue* found = nullptr;
        for (const auto& kv : obj) if (kv.first == key) found = &kv.second;
        return found;
    }
    // Option<T> semantics: a missing key and an explicit null are both None
    const JsonValue* opt(const std::string& key) const { const JsonValue* v = get(key); return (v && !v->is_null()) ? v : nullptr; }
};

class JsonParser {
public:
    explicit JsonParser(const std::string& text) : s_(text) {}
    JsonValue parse() {
        JsonValue v = value();
        ws();
        if (p_ != s_.size()) err("trailing characters");
        return v;
    }

private:
    const std::string& s_;
    size_t p_ = 0;
    int depth_ = 0;
    static constexpr int MAX_DEPTH = 128;      // serde_json's default recursion limit
    struct Nest { JsonParser& p; explicit Nest(JsonParser& q) : p(q) { if (++p.depth_ > MAX_DEPTH) p.err("recursion limit exceeded"); } ~Nest() { --p.depth_; } };
    // RFC 8259 number: -? (0 | [1-9][0-9]*) (\.[0-9]+)? ([eE][+-]?[0-9]+)?   -- strtod alone would also take hex, inf, nan, "1."
    size_t number_end(bool& integral) const {
        size_t q = p_; integral = true;
        if (q < s_.size() && s_[q] == '-') { ++q; integral = false; }
        if (q >= s_.size() || !(s_[q] >= '0' && s_[q] <= '9')) return p_;
        if (s_[q] == '0') ++q; else while (q < s_.size() && s_[q] >= '0' && s_[q] <= '9') ++q;
        if (q < s_.size() && s_[q] == '.') {
            size_t d = ++q; while (q < s_.size() && s_[q] >= '0' && s_[q] <= '9') ++q;
            if (q == d) return p_;
            integral = false;
        }
        if (q < s_.size() && (s_[q] == 'e' || s_[q] == 'E')) {
            ++q; if (q < s_.size() && (s_[q] == '+' || s_[q] == '-')) ++q;
            size_t d = q; while (q < s_.size() && s_[q] >= '0' && s_[q] <= '9') ++q;
            if (q == d) return p_;
            integral = false;
        }
        return q;
    }
    [[noreturn]] void err(const char* what) const { throw std::runtime_error(std::string("JSON: ") + what + " at byte " + std::to_string(p_)); }
    void ws() { while (p_ < s_.size() && (s_[p_] == ' ' || s_[p_] == '\t' || s_[p_] == '\n' || s_[p_] == '\r')) ++p_; }
    bool lit(const char* w) { size_t n = std::char_traits<char>::length(w); if (s_.compare(p_, n, w) == 0) { p_ += n; return true; } return false; }
    JsonValue value() {
        ws();
        if (p_ >= s_.size()) err("unexpected end");
        char c = s_[p_];
        JsonValue v;
        if (c == '{') {
            Nest nest(*this);
            v.kind = JsonValue::Object; ++p_; ws();
            if (p_ < s_.size() && s_[p_] == '}') { ++p_; return v; }
            for (;;) {
                ws(); if (p_ >= s_.size() || s_[p_] != '"') err("expected string key");
                std::string k = string();
                ws(); if (p_ >= s_.size() || s_[p_] != ':') err("expected ':'");
                ++p_;
                v.obj.emplace_back(std::move(k), value());
                ws(); if (p_ >= s_.size()) err("unexpected end in object");
                if (s_[p_] == ',') { ++p_; continue; }
                if (s_[p_] == '}') { ++p_; break; }
                err("expected ',' or '}'");
            }
        } else if (c == '[') {
            Nest nest(*this);
            v.kind = JsonValue::Array; ++p_; ws();
            if (p_ < s_.size() && s_[p_] == ']') { ++p_; return v; }
            for (;;) {
                v.arr.push_back(value());
                ws(); if (p_ >= s_.size()) err("unexpected end in array");
                if (s_[p_] == ',') { ++p_; continue; }
                if (s_[p_] == ']') { ++p_; break; }
                err("expected ',' or ']'");
            }
        } else if (c == '"') {
            v.kind = JsonValue::String; v.str = string();
        } else if (lit("true")) { v.kind = JsonValue::Bool; v.b = true;
        } else if (lit("false")) { v.kind = JsonValue::Bool; v.b = false;
        } else if (lit("null")) { v.kind = JsonValue::Null;
        } else if (c == '-' || (c >= '0' && c <= '9')) {
            const size_t q = number_end(v.integral);
            if (q == p_) err("bad number");
            v.kind = JsonValue::Number; v.num = std::strtod(s_.substr(p_, q - p_).c_str(), nullptr);
            p_ = q;
        } else err("unexpected character");
        return v;
    }
    std::string string() {
        std::string out; ++p_;
        while (p_ < s_.size() && s_[p_] != '"') {
            char c = s_[p_++];
            if (c != '\\') { out.push_back(c); continue; }
            if (p_ >= s_.size()) err("bad escape");
            char e = s_[p_++];
            switch (e) {
                case '"': out.push_back('"'); break; case '\\': out.push_back('\\'); break; case '/': out.push_back('/'); break;
                case 'b': out.push_back('\b'); break; case 'f': out.push_back('\f'); break; case 'n': out.push_back('\n'); break;
                case 'r': out.push_back('\r'); break; case 't': out.push_back('\t'); break;
                case 'u': {
                    if (p_ + 4 > s_.size()) err("bad \\u escape");
                    unsigned cp = (unsigned)std::strtoul(s_.substr(p_, 4).c_str(), nullptr, 16); p_ += 4;
                    if (cp < 0x80) out.push_back((char)cp);
                    else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
                    else { out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
                    break;
                }
                default: err("bad escape");
            }
        }
        if (p_ >= s_.size()) err("unterminated string");
        ++p_;
        return out;
    }
};

}  // namespace mi355rt_host
