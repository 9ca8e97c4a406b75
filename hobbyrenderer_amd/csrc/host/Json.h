// Json.h -- a small JSON document model for the glTF / .scene readers (the reference uses cgltf's jsmn tokens and a second jsmn
// pass for its own scene files, src/SceneLoader.cpp:50-160). Strict RFC 8259 parsing, UTF-8 strings, \uXXXX escapes with surrogate
// pairs, numbers kept as double plus the exact integer when the literal is one.
#pragma once

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace hobbyrt {
namespace json {

struct Value {
    enum Type { Null, Bool, Number, String, Array, Object };
    Type type = Null;
    bool boolean = false;
    double number = 0.0;
    bool isInteger = false; int64_t integer = 0;
    float single = 0.0f;        // the literal converted straight to binary32 (strtof), for readers that use std::stof on the token text
    std::string string;
    std::vector<Value> array;
    std::vector<std::pair<std::string, Value>> object;   // insertion order kept

    bool is(Type t) const { return type == t; }
    const Value* find(const char* key) const;             // object member or nullptr
    const Value& operator[](const char* key) const;       // object member or a shared Null value
    const Value& operator[](size_t i) const;              // array element or a shared Null value
    const Value& operator[](int i) const { return (*this)[(size_t)(i < 0 ? ~(size_t)0 : (size_t)i)]; }
    size_t size() const { return type == Array ? array.size() : (type == Object ? object.size() : 0); }
    // typed reads with defaults (the glTF schema's "default" column)
    double num(double dflt) const { return type == Number ? number : dflt; }
    float f32(float dflt) const { return type == Number ? (float)number : dflt; }          // via double, as cgltf does
    float f32_direct(float dflt) const { return type == Number ? single : dflt; }         // as std::stof would
    int64_t i64(int64_t dflt) const { return type == Number ? (isInteger ? integer : (int64_t)number) : dflt; }
    bool flag(bool dflt) const { return type == Bool ? boolean : dflt; }
    std::string str(const std::string& dflt) const { return type == String ? string : dflt; }   // by value: callers bind it to references
};

// Parses `n` bytes; on failure returns false and sets err to "offset N: reason".
bool parse(const char* text, size_t n, Value& out, std::string& err);

} // namespace json
} // namespace hobbyrt
