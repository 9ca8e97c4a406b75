#include "Json.h"

#include <cmath>
#include <cstdlib>
#include <cstring>

namespace hobbyrt {
namespace json {

namespace {
const Value kNull;

struct Parser {
    const char* p; const char* end; const char* begin; std::string err; int depth = 0;

    bool fail(const char* why) { if (err.empty()) err = "offset " + std::to_string((size_t)(p - begin)) + ": " + why; return false; }
    void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }

    static void utf8(std::string& s, uint32_t cp)
    {
        if (cp < 0x80) s.push_back((char)cp);
        else if (cp < 0x800) { s.push_back((char)(0xC0 | (cp >> 6))); s.push_back((char)(0x80 | (cp & 0x3F))); }
        else if (cp < 0x10000) { s.push_back((char)(0xE0 | (cp >> 12))); s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F))); }
        else { s.push_back((char)(0xF0 | (cp >> 18))); s.push_back((char)(0x80 | ((cp >> 12) & 0x3F))); s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F))); }
    }
    bool hex4(uint32_t& v)
    {
        if (end - p < 4) return fail("short \\u escape");
        v = 0;
        for (int i = 0; i < 4; ++i) {
            char c = *p++; uint32_t d;
            if (c >= '0' && c <= '9') d = (uint32_t)(c - '0'); else if (c >= 'a' && c <= 'f') d = (uint32_t)(c - 'a' + 10); else if (c >= 'A' && c <= 'F') d = (uint32_t)(c - 'A' + 10);
            else return fail("bad hex digit in \\u escape");
            v = (v << 4) | d;
        }
        return true;
    }
    bool string(std::string& out)
    {
        ++p;   // opening quote
        for (;;) {
            if (p >= end) return fail("unterminated string");
            unsigned char c = (unsigned char)*p++;
            if (c == '"') return true;
            if (c < 0x20) return fail("control character in string");
            if (c != '\\') { out.push_back((char)c); continue; }
            if (p >= end) return fail("unterminated escape");
            char e = *p++;
            switch (e) {
            case '"': out.push_back('"'); break; case '\\': out.push_back('\\'); break; case '/': out.push_back('/'); break;
            case 'b': out.push_back('\b'); break; case 'f': out.push_back('\f'); break; case 'n': out.push_back('\n'); break;
            case 'r': out.push_back('\r'); break; case 't': out.push_back('\t'); break;
            case 'u': {
                uint32_t cp = 0; if (!hex4(cp)) return false;
                if (cp >= 0xD800 && cp < 0xDC00) {          // high surrogate: a low one must follow
                    uint32_t lo = 0;
                    if (end - p < 2 || p[0] != '\\' || p[1] != 'u') return fail("lone high surrogate");
                    p += 2; if (!hex4(lo)) return false;
                    if (lo < 0xDC00 || lo > 0xDFFF) return fail("bad low surrogate");
                    cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                } else if (cp >= 0xDC00 && cp <= 0xDFFF) return fail("lone low surrogate");
                utf8(out, cp); break;
            }
            default: return fail("unknown escape");
            }
        }
    }
    bool number(Value& v)
    {
        const char* s = p;
        if (p < end && *p == '-') ++p;
        if (p >= end) return fail("bad number");
        if (*p == '0') ++p;
        else if (*p >= '1' && *p <= '9') { while (p < end && *p >= '0' && *p <= '9') ++p; }
        else return fail("bad number");
        bool integral = true;
        if (p < end && *p == '.') { integral = false; ++p; if (p >= end || *p < '0' || *p > '9') return fail("bad fraction"); while (p < end && *p >= '0' && *p <= '9') ++p; }
        if (p < end && (*p == 'e' || *p == 'E')) {
            integral = false; ++p;
            if (p < end && (*p == '+' || *p == '-')) ++p;
            if (p >= end || *p < '0' || *p > '9') return fail("bad exponent");
            while (p < end && *p >= '0' && *p <= '9') ++p;
        }
        std::string lit(s, p);
        v.type = Value::Number;
        v.number = std::strtod(lit.c_str(), nullptr);
        v.single = std::strtof(lit.c_str(), nullptr);
        if (integral && lit.size() <= 18) { v.isInteger = true; v.integer = std::strtoll(lit.c_str(), nullptr, 10); }
        return true;
    }
    bool literal(const char* word, size_t n) { if ((size_t)(end - p) < n || std::memcmp(p, word, n) != 0) return fail("bad literal"); p += n; return true; }

    bool value(Value& v)
    {
        ws();
        if (p >= end) return fail("unexpected end");
        if (++depth > 256) return fail("nesting too deep");
        bool ok = false;
        switch (*p) {
        case '{': {
            v.type = Value::Object; ++p; ws();
            if (p < end && *p == '}') { ++p; ok = true; break; }
            for (;;) {
                ws();
                if (p >= end || *p != '"') { fail("object key expected"); break; }
                std::string key; if (!string(key)) break;
                ws();
                if (p >= end || *p != ':') { fail("':' expected"); break; }
                ++p;
                v.object.emplace_back(std::move(key), Value());
                if (!value(v.object.back().second)) break;
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; ok = true; }
                else fail("',' or '}' expected");
                break;
            }
            break;
        }
        case '[': {
            v.type = Value::Array; ++p; ws();
            if (p < end && *p == ']') { ++p; ok = true; break; }
            for (;;) {
                v.array.emplace_back();
                if (!value(v.array.back())) break;
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; ok = true; }
                else fail("',' or ']' expected");
                break;
            }
            break;
        }
        case '"': v.type = Value::String; ok = string(v.string); break;
        case 't': ok = literal("true", 4); v.type = Value::Bool; v.boolean = true; break;
        case 'f': ok = literal("false", 5); v.type = Value::Bool; v.boolean = false; break;
        case 'n': ok = literal("null", 4); v.type = Value::Null; break;
        default: ok = number(v); break;
        }
        --depth;
        return ok;
    }
};
} // namespace

const Value* Value::find(const char* key) const
{
    if (type != Object) return nullptr;
    for (const auto& kv : object) if (kv.first == key) return &kv.second;
    return nullptr;
}
const Value& Value::operator[](const char* key) const { const Value* v = find(key); return v ? *v : kNull; }
const Value& Value::operator[](size_t i) const { return (type == Array && i < array.size()) ? array[i] : kNull; }

bool parse(const char* text, size_t n, Value& out, std::string& err)
{
    Parser ps{ text, text + n, text, std::string() };
    if (n >= 3 && (unsigned char)text[0] == 0xEF && (unsigned char)text[1] == 0xBB && (unsigned char)text[2] == 0xBF) ps.p += 3;   // UTF-8 BOM
    out = Value();
    if (!ps.value(out)) { err = ps.err; return false; }
    ps.ws();
    if (ps.p != ps.end) { ps.fail("trailing characters"); err = ps.err; return false; }
    return true;
}

} // namespace json
} // namespace hobbyrt
