// tr_error.h -- thread-local "last error" text behind tr_last_error().
#pragma once

#include <string>

namespace tr {

// Records `msg` for tr_last_error() and returns `code` (a TR_E_* value).
int fail(int code, const std::string &msg);

}  // namespace tr
