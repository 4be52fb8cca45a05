// error plumbing of the library, for the one-kernel experiment builds
#include <cstdio>
#include <string>
namespace desco {
std::string& last_error_ref() { static std::string s; return s; }
int fail(int code, const char* msg) { last_error_ref() = msg; std::fprintf(stderr, "desco: %s\n", msg); return code ? code : -1; }
}
