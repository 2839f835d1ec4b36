// json_min.hpp — a small recursive-descent JSON reader, enough for the glTF JSON chunk and
// for nif_metadata.txt. No dependency beyond the standard library.
#pragma once

#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace mi::json {

struct Value;
using ValuePtr = std::shared_ptr<Value>;

struct Value {
  enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
  bool b = false;
  double num = 0.0;
  std::string str;
  std::vector<ValuePtr> arr;
  std::map<std::string, ValuePtr> obj;

  bool has(const std::string& k) const { return kind == Object && obj.count(k); }
  const Value& at(const std::string& k) const {
    auto it = obj.find(k);
    if (kind != Object || it == obj.end()) throw std::runtime_error("json: missing key '" + k + "'");
    return *it->second;
  }
  const Value& at(size_t i) const {
    if (kind != Array || i >= arr.size()) throw std::runtime_error("json: index out of range");
    return *arr[i];
  }
  size_t size() const { return kind == Array ? arr.size() : obj.size(); }
  double number() const { if (kind != Number) throw std::runtime_error("json: not a number"); return num; }
  const std::string& string() const { if (kind != String) throw std::runtime_error("json: not a string"); return str; }
};

class Parser {
 public:
  explicit Parser(const std::string& text) : s(text) {}
  ValuePtr parse() {
    ValuePtr v = value();
    ws();
    if (p != s.size()) fail("trailing characters");
    return v;
  }

 private:
  const std::string& s;
  size_t p = 0;

  [[noreturn]] void fail(const char* what) const {
    throw std::runtime_error(std::string("json: ") + what + " at offset " + std::to_string(p));
  }
  void ws() { while (p < s.size() && (s[p] == ' ' || s[p] == '\t' || s[p] == '\n' || s[p] == '\r')) ++p; }
  bool eat(char c) { ws(); if (p < s.size() && s[p] == c) { ++p; return true; } return false; }

  ValuePtr value() {
    ws();
    if (p >= s.size()) fail("unexpected end");
    auto v = std::make_shared<Value>();
    const char c = s[p];
    if (c == '{') {
      ++p; v->kind = Value::Object;
      if (eat('}')) return v;
      do {
        ws();
        if (p >= s.size() || s[p] != '"') fail("expected key");
        std::string k = quoted();
        if (!eat(':')) fail("expected ':'");
        v->obj[k] = value();
      } while (eat(','));
      if (!eat('}')) fail("expected '}'");
    } else if (c == '[') {
      ++p; v->kind = Value::Array;
      if (eat(']')) return v;
      do { v->arr.push_back(value()); } while (eat(','));
      if (!eat(']')) fail("expected ']'");
    } else if (c == '"') {
      v->kind = Value::String; v->str = quoted();
    } else if (s.compare(p, 4, "true") == 0) { p += 4; v->kind = Value::Bool; v->b = true; }
    else if (s.compare(p, 5, "false") == 0) { p += 5; v->kind = Value::Bool; v->b = false; }
    else if (s.compare(p, 4, "null") == 0) { p += 4; v->kind = Value::Null; }
    else {
      const char* start = s.c_str() + p;
      char* end = nullptr;
      v->num = std::strtod(start, &end);
      if (end == start) fail("bad token");
      v->kind = Value::Number;
      p += (size_t)(end - start);
    }
    return v;
  }

  std::string quoted() {
    std::string out;
    ++p;  // opening quote
    while (p < s.size() && s[p] != '"') {
      char c = s[p++];
      if (c == '\\' && p < s.size()) {
        char e = s[p++];
        switch (e) {
          case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break;
          case 'b': out += '\b'; break; case 'f': out += '\f'; break;
          case 'u': out += '?'; p += 4; break;   // non-ASCII escapes are not needed by our inputs
          default: out += e;
        }
      } else out += c;
    }
    if (p >= s.size()) fail("unterminated string");
    ++p;
    return out;
  }
};

inline ValuePtr parse(const std::string& text) { return Parser(text).parse(); }

}  // namespace mi::json
