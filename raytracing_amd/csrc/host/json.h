// Minimal JSON reader for glTF 2.0 (the reference uses nlohmann::json, model.h:3, which is not in this
// image).  Supports exactly what model.cpp touches: operator[] by key / index, find-like has(),
// value(key, default), size(), number / string conversion.
#pragma once
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace rt_host {

class Json {
public:
    enum Type { Null, Bool, Number, String, Array, Object };
    Type type = Null;
    bool b = false;
    double num = 0.0;
    bool isInteger = false;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;   // insertion order kept

    static Json parse(const std::string& text) {
        size_t i = 0;
        Json j = parseValue(text, i);
        skipWs(text, i);
        if (i != text.size()) throw std::runtime_error("json: trailing characters");
        return j;
    }
    bool has(const std::string& k) const {
        if (type != Object) return false;
        for (auto& kv : obj) if (kv.first == k) return true;
        return false;
    }
    const Json& operator[](const std::string& k) const {
        if (type == Object) for (auto& kv : obj) if (kv.first == k) return kv.second;
        throw std::runtime_error("json: missing key '" + k + "'");
    }
    const Json& operator[](size_t i) const {
        if (type != Array || i >= arr.size()) throw std::runtime_error("json: array index out of range");
        return arr[i];
    }
    size_t size() const { return type == Array ? arr.size() : (type == Object ? obj.size() : (type == Null ? 0 : 1)); }
    double number() const { if (type != Number) throw std::runtime_error("json: not a number"); return num; }
    const std::string& string() const { if (type != String) throw std::runtime_error("json: not a string"); return str; }
    double value(const std::string& k, double dflt) const { return has(k) ? (*this)[k].number() : dflt; }

private:
    static void skipWs(const std::string& s, size_t& i) { while (i < s.size() && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r')) i++; }
    static Json parseValue(const std::string& s, size_t& i) {
        skipWs(s, i);
        if (i >= s.size()) throw std::runtime_error("json: unexpected end");
        char c = s[i];
        Json j;
        if (c == '{') {
            j.type = Object; i++;
            skipWs(s, i);
            if (i < s.size() && s[i] == '}') { i++; return j; }
            while (true) {
                skipWs(s, i);
                std::string k = parseString(s, i);
                skipWs(s, i);
                if (i >= s.size() || s[i] != ':') throw std::runtime_error("json: expected ':'");
                i++;
                j.obj.emplace_back(k, parseValue(s, i));
                skipWs(s, i);
                if (i < s.size() && s[i] == ',') { i++; continue; }
                if (i < s.size() && s[i] == '}') { i++; break; }
                throw std::runtime_error("json: expected ',' or '}'");
            }
        } else if (c == '[') {
            j.type = Array; i++;
            skipWs(s, i);
            if (i < s.size() && s[i] == ']') { i++; return j; }
            while (true) {
                j.arr.push_back(parseValue(s, i));
                skipWs(s, i);
                if (i < s.size() && s[i] == ',') { i++; continue; }
                if (i < s.size() && s[i] == ']') { i++; break; }
                throw std::runtime_error("json: expected ',' or ']'");
            }
        } else if (c == '"') {
            j.type = String; j.str = parseString(s, i);
        } else if (s.compare(i, 4, "true") == 0) { j.type = Bool; j.b = true; i += 4;
        } else if (s.compare(i, 5, "false") == 0) { j.type = Bool; j.b = false; i += 5;
        } else if (s.compare(i, 4, "null") == 0) { j.type = Null; i += 4;
        } else {
            size_t st = i;
            bool integer = true;
            while (i < s.size() && (std::isdigit((unsigned char)s[i]) || s[i] == '-' || s[i] == '+' || s[i] == '.' || s[i] == 'e' || s[i] == 'E')) {
                if (s[i] == '.' || s[i] == 'e' || s[i] == 'E') integer = false;
                i++;
            }
            if (st == i) throw std::runtime_error("json: bad token");
            j.type = Number; j.num = std::strtod(s.substr(st, i - st).c_str(), nullptr); j.isInteger = integer;
        }
        return j;
    }
    static std::string parseString(const std::string& s, size_t& i) {
        if (i >= s.size() || s[i] != '"') throw std::runtime_error("json: expected string");
        i++;
        std::string out;
        while (i < s.size() && s[i] != '"') {
            if (s[i] == '\\' && i + 1 < s.size()) {
                char e = s[i + 1];
                i += 2;
                switch (e) {
                    case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break; case 'f': out += '\f'; break;
                    case 'u': { unsigned cp = (unsigned)std::strtoul(s.substr(i, 4).c_str(), nullptr, 16); i += 4;
                                if (cp < 0x80) out += (char)cp; else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
                                else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); } break; }
                    default: out += e;
                }
            } else out += s[i++];
        }
        if (i >= s.size()) throw std::runtime_error("json: unterminated string");
        i++;
        return out;
    }
};

}  // namespace rt_host
