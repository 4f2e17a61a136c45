// Laboratory builds only (-DGLASS_LAB=1): the knobs the product build treats as constants (glass_amd/csrc/dense_common.h
// lab_knob) are read from the environment here, once per name per process.  Never linked into glass_amd/libglass_hip.so.
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

namespace glass {

int lab_knob(const char* name, int dflt) {
    static std::mutex mu;
    static std::map<std::string, int> seen;
    std::lock_guard<std::mutex> lock(mu);
    auto it = seen.find(name);
    if (it != seen.end()) return it->second;
    const char* e = std::getenv(name);
    const int v = (e && *e) ? std::atoi(e) : dflt;
    seen.emplace(name, v);
    return v;
}

}  // namespace glass
