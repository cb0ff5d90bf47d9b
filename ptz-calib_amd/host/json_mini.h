// json_mini.h -- the small part of JSON the reference's camera files need (data_io.cc:112-295 uses nlohmann::json, which
// is not available to this build): a value tree, a strict parser, and a writer that lays text out like
// nlohmann's dump(4) -- insertion-ordered objects, 4-space indentation, shortest round-trip numbers with a
// trailing ".0" on integral floating-point values -- so that files written here diff cleanly against the reference's.
#pragma once

#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace ptzcalib {

class Json {
 public:
  enum Type { kNull, kBool, kInt, kFloat, kString, kArray, kObject };
  Json() = default;
  static Json Null() { return Json(); }
  static Json Bool(bool b) { Json j; j.type_ = kBool; j.b_ = b; return j; }
  static Json Int(long long v) { Json j; j.type_ = kInt; j.i_ = v; return j; }
  static Json Float(double v) { Json j; j.type_ = kFloat; j.d_ = v; return j; }
  static Json String(const std::string& s) { Json j; j.type_ = kString; j.s_ = s; return j; }
  static Json Array() { Json j; j.type_ = kArray; return j; }
  static Json Object() { Json j; j.type_ = kObject; return j; }
  static Json FloatArray(const std::vector<double>& v);

  Type type() const { return type_; }
  bool is_object() const { return type_ == kObject; }
  bool is_array() const { return type_ == kArray; }
  bool is_number() const { return type_ == kInt || type_ == kFloat; }
  bool empty() const { return type_ == kNull || (type_ == kObject && members_.empty()) || (type_ == kArray && items_.empty()); }

  // accessors throw std::runtime_error on a type mismatch or a missing key (nlohmann throws as well; the callers catch)
  double number() const;
  long long integer() const;
  const std::string& string() const;
  const std::vector<Json>& items() const;
  bool contains(const std::string& key) const;
  const Json& at(const std::string& key) const;
  const Json& at(size_t i) const;
  std::vector<double> number_array() const;
  // object members in insertion order / in sorted key order (nlohmann::json, unlike ordered_json, iterates sorted)
  const std::vector<std::pair<std::string, Json>>& members() const;
  std::vector<std::string> sorted_keys() const;

  Json& operator[](const std::string& key);  // creates the member (and turns a null value into an object)
  void push_back(const Json& v);             // turns a null value into an array

  std::string dump(int indent = 4) const;
  static bool Parse(const std::string& text, Json& out, std::string* error = nullptr);

 private:
  void DumpTo(std::string& out, int indent, int level) const;
  Type type_ = kNull;
  bool b_ = false;
  long long i_ = 0;
  double d_ = 0;
  std::string s_;
  std::vector<Json> items_;
  std::vector<std::pair<std::string, Json>> members_;
};

}  // namespace ptzcalib
