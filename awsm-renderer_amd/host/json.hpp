// json.hpp — minimal JSON DOM for the glTF reader (RFC 8259: objects, arrays, strings with escapes, numbers, true/false/null).
// No external dependency is available in this image; the reference relies on the `gltf` crate (serde_json) for the same job.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace awsm_json {

struct Value {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<Value> arr;
    std::vector<std::pair<std::string, Value>> obj;   // insertion order kept

    bool is_null() const { return kind == Null; }
    bool is_object() const { return kind == Object; }
    bool is_array() const { return kind == Array; }
    bool is_number() const { return kind == Number; }
    bool is_string() const { return kind == String; }
    size_t size() const { return kind == Array ? arr.size() : (kind == Object ? obj.size() : 0); }
    const Value& operator[](size_t i) const { static const Value none; return kind == Array && i < arr.size() ? arr[i] : none; }
    const Value& operator[](const char* key) const {
        static const Value none;
        if (kind != Object) return none;
        for (const auto& kv : obj) if (kv.first == key) return kv.second;
        return none;
    }
    bool has(const char* key) const { return !(*this)[key].is_null(); }
    double number(double dflt) const { return kind == Number ? num : dflt; }
    int64_t integer(int64_t dflt) const {      // non-finite or beyond +-2^62: the default (llround of those is undefined)
        return kind == Number && num >= -4.0e18 && num <= 4.0e18 ? (int64_t)std::llround(num) : dflt;
    }
    bool boolean(bool dflt) const { return kind == Bool ? b : dflt; }
    const std::string& string() const { return str; }
};

class Parser {
public:
    Parser(const char* text, size_t len) : p_(text), end_(text + len) {}
    bool parse(Value& out, std::string& err) {
        skip();
        if (!value(out, 0)) { err = err_.empty() ? "syntax error" : err_; return false; }
        skip();
        if (p_ != end_) { err = "trailing characters after the JSON value"; return false; }
        return true;
    }

private:
    const char* p_;
    const char* end_;
    std::string err_;

    void skip() { while (p_ < end_ && (*p_ == ' ' || *p_ == '\t' || *p_ == '\n' || *p_ == '\r')) p_++; }
    bool fail(const char* m) { if (err_.empty()) err_ = m; return false; }
    bool lit(const char* s) { size_t n = strlen(s); if ((size_t)(end_ - p_) < n || memcmp(p_, s, n) != 0) return false; p_ += n; return true; }

    static void utf8(std::string& s, uint32_t c) {
        if (c < 0x80) s += (char)c;
        else if (c < 0x800) { s += (char)(0xC0 | (c >> 6)); s += (char)(0x80 | (c & 0x3F)); }
        else if (c < 0x10000) { s += (char)(0xE0 | (c >> 12)); s += (char)(0x80 | ((c >> 6) & 0x3F)); s += (char)(0x80 | (c & 0x3F)); }
        else { s += (char)(0xF0 | (c >> 18)); s += (char)(0x80 | ((c >> 12) & 0x3F)); s += (char)(0x80 | ((c >> 6) & 0x3F)); s += (char)(0x80 | (c & 0x3F)); }
    }
    bool hex4(uint32_t& v) {
        if (end_ - p_ < 4) return false;
        v = 0;
        for (int i = 0; i < 4; i++) {
            char c = *p_++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
            else return false;
        }
        return true;
    }
    bool string(std::string& s) {
        if (p_ >= end_ || *p_ != '"') return fail("expected a string");
        p_++;
        while (p_ < end_ && *p_ != '"') {
            char c = *p_++;
            if (c != '\\') { s += c; continue; }
            if (p_ >= end_) return fail("unterminated escape");
            char e = *p_++;
            switch (e) {
                case '"': s += '"'; break; case '\\': s += '\\'; break; case '/': s += '/'; break;
                case 'b': s += '\b'; break; case 'f': s += '\f'; break; case 'n': s += '\n'; break; case 'r': s += '\r'; break; case 't': s += '\t'; break;
                case 'u': {
                    uint32_t c1;
                    if (!hex4(c1)) return fail("bad \\u escape");
                    if (c1 >= 0xD800 && c1 < 0xDC00 && end_ - p_ >= 6 && p_[0] == '\\' && p_[1] == 'u') {
                        p_ += 2;
                        uint32_t c2;
                        if (!hex4(c2)) return fail("bad \\u escape");
                        c1 = 0x10000 + ((c1 - 0xD800) << 10) + (c2 - 0xDC00);
                    }
                    utf8(s, c1);
                    break;
                }
                default: return fail("unknown escape");
            }
        }
        if (p_ >= end_) return fail("unterminated string");
        p_++;
        return true;
    }
    bool value(Value& v, int depth) {
        if (depth > 256) return fail("nesting too deep");
        skip();
        if (p_ >= end_) return fail("unexpected end of input");
        char c = *p_;
        if (c == '{') {
            p_++; v.kind = Value::Object; skip();
            if (p_ < end_ && *p_ == '}') { p_++; return true; }
            for (;;) {
                skip();
                std::string key;
                if (!string(key)) return false;
                skip();
                if (p_ >= end_ || *p_ != ':') return fail("expected ':'");
                p_++;
                v.obj.emplace_back(std::move(key), Value());
                if (!value(v.obj.back().second, depth + 1)) return false;
                skip();
                if (p_ < end_ && *p_ == ',') { p_++; continue; }
                if (p_ < end_ && *p_ == '}') { p_++; return true; }
                return fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            p_++; v.kind = Value::Array; skip();
            if (p_ < end_ && *p_ == ']') { p_++; return true; }
            for (;;) {
                v.arr.emplace_back();
                if (!value(v.arr.back(), depth + 1)) return false;
                skip();
                if (p_ < end_ && *p_ == ',') { p_++; continue; }
                if (p_ < end_ && *p_ == ']') { p_++; return true; }
                return fail("expected ',' or ']'");
            }
        }
        if (c == '"') { v.kind = Value::String; return string(v.str); }
        if (lit("true")) { v.kind = Value::Bool; v.b = true; return true; }
        if (lit("false")) { v.kind = Value::Bool; v.b = false; return true; }
        if (lit("null")) { v.kind = Value::Null; return true; }
        if (c == '-' || (c >= '0' && c <= '9')) {
            const char* s = p_;
            if (*p_ == '-') p_++;
            while (p_ < end_ && ((*p_ >= '0' && *p_ <= '9') || *p_ == '.' || *p_ == 'e' || *p_ == 'E' || *p_ == '+' || *p_ == '-')) p_++;
            std::string tmp(s, (size_t)(p_ - s));
            char* endp = nullptr;
            v.num = strtod(tmp.c_str(), &endp);
            if (!endp || *endp) return fail("bad number");
            v.kind = Value::Number;
            return true;
        }
        return fail("unexpected character");
    }
};

inline bool parse(const char* text, size_t len, Value& out, std::string& err) { return Parser(text, len).parse(out, err); }

}  // namespace awsm_json
