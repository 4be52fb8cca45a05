// Shared error plumbing of libdesco_hip.so (host side).
#pragma once
#include <string>

namespace desco {
std::string& last_error_ref();
int fail(int code, const char* msg);
}  // namespace desco
