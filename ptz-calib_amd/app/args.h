// args.h -- the handful of command-line conventions of the reference's cmdline.h parser that its two tools use:
// --name value, --name=value, -x value, boolean --flag; a missing required option or an unknown option prints the usage
// text and exits with status 1 (cmdline.h:595-598).
#pragma once

#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

namespace ptzapp {

struct Option { std::string name; char short_name; std::string desc; bool required; bool is_flag; };

class Args {
 public:
  void Add(const std::string& name, char short_name, const std::string& desc, bool required) { opts_.push_back({name, short_name, desc, required, false}); }
  void AddFlag(const std::string& name, const std::string& desc) { opts_.push_back({name, '\0', desc, false, true}); }
  void ParseCheck(int argc, char** argv)
  {
    prog_ = argc > 0 ? argv[0] : "prog";
    for (int i = 1; i < argc; ++i) {
      std::string a = argv[i], key, val;
      bool has_val = false;
      const Option* o = nullptr;
      if (a.rfind("--", 0) == 0) {
        key = a.substr(2);
        const size_t eq = key.find('=');
        if (eq != std::string::npos) { val = key.substr(eq + 1); key = key.substr(0, eq); has_val = true; }
        for (const Option& c : opts_) if (c.name == key) o = &c;
      }
      else if (a.size() == 2 && a[0] == '-') {
        for (const Option& c : opts_) if (c.short_name && c.short_name == a[1]) o = &c;
      }
      if (!o) Fail("undefined option: " + a);
      if (o->is_flag) { flags_[o->name] = true; continue; }
      if (!has_val) {
        if (i + 1 >= argc) Fail("option needs value: --" + o->name);
        val = argv[++i];
      }
      values_[o->name] = val;
    }
    for (const Option& c : opts_)
      if (c.required && !values_.count(c.name)) Fail("need option: --" + c.name);
  }
  std::string Get(const std::string& name) const { auto it = values_.find(name); return it == values_.end() ? "" : it->second; }
  bool Exist(const std::string& name) const { return flags_.count(name) != 0 || values_.count(name) != 0; }

 private:
  [[noreturn]] void Fail(const std::string& msg) const
  {
    fprintf(stderr, "%s\nusage: %s", msg.c_str(), prog_.c_str());
    for (const Option& c : opts_) fprintf(stderr, c.is_flag ? " [--%s]" : (c.required ? " --%s=string" : " [--%s=string]"), c.name.c_str());
    fprintf(stderr, "\noptions:\n");
    for (const Option& c : opts_) {
      if (c.short_name) fprintf(stderr, "  -%c, --%-16s %s\n", c.short_name, c.name.c_str(), c.desc.c_str());
      else fprintf(stderr, "      --%-16s %s\n", c.name.c_str(), c.desc.c_str());
    }
    exit(1);
  }
  std::vector<Option> opts_;
  std::map<std::string, std::string> values_;
  std::map<std::string, bool> flags_;
  std::string prog_;
};

}  // namespace ptzapp
