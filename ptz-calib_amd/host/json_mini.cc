#include "json_mini.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace ptzcalib {

Json Json::FloatArray(const std::vector<double>& v)
{
  Json j = Array();
  for (double x : v) j.items_.push_back(Float(x));
  return j;
}

double Json::number() const
{
  if (type_ == kFloat) return d_;
  if (type_ == kInt) return static_cast<double>(i_);
  throw std::runtime_error("json: value is not a number");
}
long long Json::integer() const
{
  if (type_ == kInt) return i_;
  if (type_ == kFloat) return static_cast<long long>(d_);  // nlohmann converts the same way for get<int>()
  throw std::runtime_error("json: value is not a number");
}
const std::string& Json::string() const
{
  if (type_ != kString) throw std::runtime_error("json: value is not a string");
  return s_;
}
const std::vector<Json>& Json::items() const
{
  if (type_ != kArray) throw std::runtime_error("json: value is not an array");
  return items_;
}
bool Json::contains(const std::string& key) const
{
  if (type_ != kObject) return false;
  for (const auto& m : members_) if (m.first == key) return true;
  return false;
}
const Json& Json::at(const std::string& key) const
{
  if (type_ != kObject) throw std::runtime_error("json: value is not an object");
  for (const auto& m : members_) if (m.first == key) return m.second;
  throw std::runtime_error("json: key '" + key + "' not found");
}
const Json& Json::at(size_t i) const
{
  if (type_ != kArray || i >= items_.size()) throw std::runtime_error("json: array index out of range");
  return items_[i];
}
std::vector<double> Json::number_array() const
{
  std::vector<double> out;
  for (const Json& v : items()) out.push_back(v.number());
  return out;
}
const std::vector<std::pair<std::string, Json>>& Json::members() const
{
  if (type_ != kObject) throw std::runtime_error("json: value is not an object");
  return members_;
}
std::vector<std::string> Json::sorted_keys() const
{
  std::vector<std::string> keys;
  for (const auto& m : members()) keys.push_back(m.first);
  std::sort(keys.begin(), keys.end());
  return keys;
}
Json& Json::operator[](const std::string& key)
{
  if (type_ == kNull) type_ = kObject;
  if (type_ != kObject) throw std::runtime_error("json: value is not an object");
  for (auto& m : members_) if (m.first == key) return m.second;
  members_.emplace_back(key, Json());
  return members_.back().second;
}
void Json::push_back(const Json& v)
{
  if (type_ == kNull) type_ = kArray;
  if (type_ != kArray) throw std::runtime_error("json: value is not an array");
  items_.push_back(v);
}

// ---- writer -----------------------------------------------------------------------------------------------------
namespace {
void DumpString(const std::string& s, std::string& out)
{
  out.push_back('"');
  for (unsigned char c : s) {
    switch (c) {
      case '"': out += "\\\""; break;
      case '\\': out += "\\\\"; break;
      case '\b': out += "\\b"; break;
      case '\f': out += "\\f"; break;
      case '\n': out += "\\n"; break;
      case '\r': out += "\\r"; break;
      case '\t': out += "\\t"; break;
      default:
        if (c < 0x20) { char buf[8]; snprintf(buf, sizeof(buf), "\\u%04x", c); out += buf; }
        else out.push_back(static_cast<char>(c));
    }
  }
  out.push_back('"');
}
// shortest decimal text that parses back to the same double; integral values keep a ".0" (nlohmann's to_chars)
void DumpDouble(double v, std::string& out)
{
  if (!std::isfinite(v)) { out += "null"; return; }
  char buf[40];
  for (int prec = 1; prec <= 17; ++prec) {
    snprintf(buf, sizeof(buf), "%.*g", prec, v);
    if (strtod(buf, nullptr) == v) break;
  }
  std::string s = buf;
  // %g exponent form "1e+20" / "1e-07" -> "1e+20" / "1e-07" are valid JSON; add ".0" only to plain integers
  if (s.find_first_of(".eEn") == std::string::npos) s += ".0";
  out += s;
}
}  // namespace

void Json::DumpTo(std::string& out, int indent, int level) const
{
  const std::string pad(static_cast<size_t>(indent) * (level + 1), ' '), pad_end(static_cast<size_t>(indent) * level, ' ');
  switch (type_) {
    case kNull: out += "null"; break;
    case kBool: out += b_ ? "true" : "false"; break;
    case kInt: out += std::to_string(i_); break;
    case kFloat: DumpDouble(d_, out); break;
    case kString: DumpString(s_, out); break;
    case kArray:
      if (items_.empty()) { out += "[]"; break; }
      out += "[\n";
      for (size_t i = 0; i < items_.size(); ++i) {
        out += pad;
        items_[i].DumpTo(out, indent, level + 1);
        out += (i + 1 < items_.size()) ? ",\n" : "\n";
      }
      out += pad_end + "]";
      break;
    case kObject:
      if (members_.empty()) { out += "{}"; break; }
      out += "{\n";
      for (size_t i = 0; i < members_.size(); ++i) {
        out += pad;
        DumpString(members_[i].first, out);
        out += ": ";
        members_[i].second.DumpTo(out, indent, level + 1);
        out += (i + 1 < members_.size()) ? ",\n" : "\n";
      }
      out += pad_end + "}";
      break;
  }
}

std::string Json::dump(int indent) const
{
  std::string out;
  DumpTo(out, indent, 0);
  return out;
}

// ---- parser -----------------------------------------------------------------------------------------------------
namespace {
struct Parser {
  const char* p;
  const char* end;
  std::string err;
  int depth = 0;

  void Skip() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }
  bool Fail(const std::string& m) { if (err.empty()) err = m; return false; }

  bool ParseString(std::string& out)
  {
    if (p >= end || *p != '"') return Fail("expected string");
    ++p;
    while (p < end && *p != '"') {
      if (*p == '\\') {
        if (++p >= end) return Fail("bad escape");
        switch (*p) {
          case '"': out.push_back('"'); break;
          case '\\': out.push_back('\\'); break;
          case '/': out.push_back('/'); break;
          case 'b': out.push_back('\b'); break;
          case 'f': out.push_back('\f'); break;
          case 'n': out.push_back('\n'); break;
          case 'r': out.push_back('\r'); break;
          case 't': out.push_back('\t'); break;
          case 'u': {
            if (end - p < 5) return Fail("bad \\u escape");
            unsigned cp = 0;
            for (int k = 1; k <= 4; ++k) {
              const char c = p[k];
              cp <<= 4;
              if (c >= '0' && c <= '9') cp |= c - '0';
              else if (c >= 'a' && c <= 'f') cp |= c - 'a' + 10;
              else if (c >= 'A' && c <= 'F') cp |= c - 'A' + 10;
              else return Fail("bad \\u escape");
            }
            p += 4;
            if (cp < 0x80) out.push_back(static_cast<char>(cp));
            else if (cp < 0x800) { out.push_back(static_cast<char>(0xC0 | (cp >> 6))); out.push_back(static_cast<char>(0x80 | (cp & 0x3F))); }
            else { out.push_back(static_cast<char>(0xE0 | (cp >> 12))); out.push_back(static_cast<char>(0x80 | ((cp >> 6) & 0x3F))); out.push_back(static_cast<char>(0x80 | (cp & 0x3F))); }
            break;
          }
          default: return Fail("bad escape");
        }
        ++p;
      }
      else out.push_back(*p++);
    }
    if (p >= end) return Fail("unterminated string");
    ++p;
    return true;
  }

  bool ParseValue(Json& out)
  {
    if (++depth > 200) return Fail("nesting too deep");
    Skip();
    if (p >= end) return Fail("unexpected end of input");
    bool ok = true;
    if (*p == '{') {
      ++p;
      out = Json::Object();
      Skip();
      if (p < end && *p == '}') { ++p; }
      else {
        while (true) {
          Skip();
          std::string key;
          if (!ParseString(key)) { ok = false; break; }
          Skip();
          if (p >= end || *p != ':') { ok = Fail("expected ':'"); break; }
          ++p;
          Json v;
          if (!ParseValue(v)) { ok = false; break; }
          out[key] = v;  // a repeated key keeps the last value, as nlohmann does
          Skip();
          if (p < end && *p == ',') { ++p; continue; }
          if (p < end && *p == '}') { ++p; break; }
          ok = Fail("expected ',' or '}'");
          break;
        }
      }
    }
    else if (*p == '[') {
      ++p;
      out = Json::Array();
      Skip();
      if (p < end && *p == ']') { ++p; }
      else {
        while (true) {
          Json v;
          if (!ParseValue(v)) { ok = false; break; }
          out.push_back(v);
          Skip();
          if (p < end && *p == ',') { ++p; continue; }
          if (p < end && *p == ']') { ++p; break; }
          ok = Fail("expected ',' or ']'");
          break;
        }
      }
    }
    else if (*p == '"') {
      std::string s;
      ok = ParseString(s);
      if (ok) out = Json::String(s);
    }
    else if (end - p >= 4 && !strncmp(p, "true", 4)) { p += 4; out = Json::Bool(true); }
    else if (end - p >= 5 && !strncmp(p, "false", 5)) { p += 5; out = Json::Bool(false); }
    else if (end - p >= 4 && !strncmp(p, "null", 4)) { p += 4; out = Json::Null(); }
    else if (*p == '-' || (*p >= '0' && *p <= '9')) {
      const char* s = p;
      bool is_float = false;
      if (*p == '-') ++p;
      if (p >= end || *p < '0' || *p > '9') { ok = Fail("bad number"); }
      else {
        while (p < end && *p >= '0' && *p <= '9') ++p;
        if (p < end && *p == '.') { is_float = true; ++p; while (p < end && *p >= '0' && *p <= '9') ++p; }
        if (p < end && (*p == 'e' || *p == 'E')) {
          is_float = true; ++p;
          if (p < end && (*p == '+' || *p == '-')) ++p;
          while (p < end && *p >= '0' && *p <= '9') ++p;
        }
        const std::string tok(s, p);
        if (is_float || tok.size() > 18) out = Json::Float(strtod(tok.c_str(), nullptr));
        else out = Json::Int(strtoll(tok.c_str(), nullptr, 10));
      }
    }
    else ok = Fail(std::string("unexpected character '") + *p + "'");
    --depth;
    return ok;
  }
};
}  // namespace

bool Json::Parse(const std::string& text, Json& out, std::string* error)
{
  Parser ps{text.data(), text.data() + text.size(), {}};
  Json v;
  bool ok = ps.ParseValue(v);
  if (ok) {
    ps.Skip();
    if (ps.p != ps.end) ok = ps.Fail("trailing characters");
  }
  if (!ok) { if (error) *error = ps.err; return false; }
  out = v;
  return true;
}

}  // namespace ptzcalib
