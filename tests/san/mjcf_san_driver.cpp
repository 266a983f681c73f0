// Sanitizer driver of the MJCF compiler behind rsim_model_compile (robosuite_amd/csrc/rsim_mjcf.cpp: plain host C++, no HIP).  Test infrastructure, built by
// tests/test_mjcf_sanitizers.py with -fsanitize=address,undefined.  For every MJCF file named on the command line: compile it as it is, then damaged copies of it --
// truncations at evenly spaced cut points, single-byte substitutions, deleted and duplicated spans (deterministic LCG) -- through the C-ABI entry rsim_mjcf_to_blob.
// A compile may succeed or fail with a reason; what must not happen is a read or write out of bounds, a use after free, a leak, signed overflow, a misaligned or null
// access -- the sanitizers abort the process on those, and the test reads the exit code.  Reference entry this replaces: mujoco.MjModel.from_xml_string
// (utils/binding_utils.py:1077-1080), which the reference trusts MuJoCo to harden.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>

struct rsim_model;
extern "C" int rsim_mjcf_to_blob(const char* xml, size_t len, const char* asset_dir, void** blob, size_t* blob_len);
extern "C" void rsim_blob_free(void* blob);
static std::string g_err;
// the two symbols rsim_mjcf.cpp takes from rsim_api.cpp (which needs the HIP runtime): the error slot, and the blob ingest behind rsim_model_compile (not driven here)
extern "C" int rsim_set_error(const char* msg) { g_err = msg ? msg : ""; return 1; }
extern "C" int rsim_model_create(const void*, size_t, rsim_model**) { return rsim_set_error("rsim_model_create: not part of the sanitizer driver"); }

static uint64_t lcg(uint64_t& s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return s >> 33; }
static int run(const std::string& xml, const char* dir, long& ok, long& bad) {
  void* blob = nullptr; size_t n = 0;
  g_err.clear();
  const int rc = rsim_mjcf_to_blob(xml.data(), xml.size(), dir, &blob, &n);
  if (rc == 0) { if (!blob || n < 16) { fprintf(stderr, "success without a blob\n"); return 2; } rsim_blob_free(blob); ok++; }
  else { if (g_err.size() < 10) { fprintf(stderr, "failure without a reason\n"); return 2; } bad++; }
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s <cuts> <mutations> file.xml[@asset_dir] ...\n", argv[0]); return 2; }
  const int cuts = atoi(argv[1]), muts = atoi(argv[2]);
  long ok = 0, bad = 0, intact_ok = 0;
  for (int a = 3; a < argc; a++) {
    std::string spec = argv[a], dir;
    const size_t at = spec.find('@');
    if (at != std::string::npos) { dir = spec.substr(at + 1); spec = spec.substr(0, at); }
    std::ifstream f(spec, std::ios::binary);
    if (!f) { fprintf(stderr, "cannot read %s\n", spec.c_str()); return 2; }
    std::stringstream ss; ss << f.rdbuf();
    const std::string xml = ss.str();
    const char* d = dir.empty() ? nullptr : dir.c_str();
    long o0 = ok;
    if (run(xml, d, ok, bad)) return 2;
    intact_ok += ok - o0;
    for (int c = 1; c <= cuts; c++) if (run(xml.substr(0, xml.size() * (size_t)c / (size_t)(cuts + 1)), d, ok, bad)) return 2;
    uint64_t s = 0x9e3779b97f4a7c15ull ^ (uint64_t)xml.size();
    static const char subst[] = "<>/\"'= \t\n0-9.eE+xX&;#\0\xff";
    for (int k = 0; k < muts && !xml.empty(); k++) {
      std::string y = xml;
      const size_t p = lcg(s) % y.size(), span = 1 + lcg(s) % 24;
      switch (lcg(s) % 4) {
        case 0: y[p] = subst[lcg(s) % (sizeof(subst) - 1)]; break;                          // one byte replaced (markup, digits, NUL, 0xff)
        case 1: y.erase(p, span); break;                                                   // a span deleted
        case 2: y.insert(p, y.substr(p, span)); break;                                     // a span duplicated
        default: { const size_t q = lcg(s) % y.size(); std::swap(y[p], y[q]); } break;     // two bytes swapped
      }
      if (run(y, d, ok, bad)) return 2;
    }
  }
  printf("compiles ok %ld (intact files: %ld of %d), rejected with a reason %ld\n", ok, intact_ok, argc - 3, bad);
  return 0;
}
