// utils.h -- Boost-free restatement of the reference's key=value ConfigFile (reference src/utils.h:282-386,
// src/utils.cc:133-197): '#' / '%' comment lines, all whitespace stripped, case-insensitive keys, typed get<T>.
#ifndef PHOTOBUNDLE_AMD_UTILS_H
#define PHOTOBUNDLE_AMD_UTILS_H

#include <algorithm>
#include <cctype>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>

namespace utils {

class ConfigFile {
 public:
  ConfigFile() {}
  explicit ConfigFile(const std::string& filename) {
    std::ifstream ifs(filename);
    if (!ifs.is_open()) throw std::runtime_error("could not open file '" + filename + "'");
    parse(ifs);
  }
  ConfigFile& operator()(const std::string& key, const std::string& value) { _data[lower(key)] = value; return *this; }

  template <typename T>
  T get(const std::string& name) const {
    auto it = _data.find(lower(name));
    if (it == _data.end()) throw std::runtime_error("no key " + name);
    return convert<T>(it->second);
  }
  template <typename T>
  T get(const std::string& name, const T& default_val) const {
    auto it = _data.find(lower(name));
    return it == _data.end() ? default_val : convert<T>(it->second);
  }
  bool has(const std::string& name) const { return _data.count(lower(name)) != 0; }

 private:
  std::map<std::string, std::string> _data;
  static std::string lower(std::string s) { std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return std::tolower(c); }); return s; }
  template <typename T>
  static T convert(const std::string& s) { std::istringstream iss(s); T v; iss >> v; if (iss.fail()) throw std::runtime_error("bad value '" + s + "'"); return v; }
  void parse(std::ifstream& ifs) {
    std::string line;
    while (std::getline(ifs, line)) {
      if (line.empty() || line.front() == '#' || line.front() == '%') continue;
      line.erase(std::remove_if(line.begin(), line.end(), [](unsigned char c) { return std::isspace(c); }), line.end());
      if (line.empty()) continue;
      const auto eq = line.find('=');
      if (eq == std::string::npos || line.find('=', eq + 1) != std::string::npos) throw std::runtime_error("Malformed ConfigFile line " + line);
      _data[lower(line.substr(0, eq))] = line.substr(eq + 1);
    }
  }
};
template <> inline std::string ConfigFile::convert<std::string>(const std::string& s) { return s; }

}  // namespace utils

#endif
