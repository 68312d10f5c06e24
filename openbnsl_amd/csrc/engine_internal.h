// Shared between the translation units of libobnsl_engine.so; not installed.
#pragma once
#include <string>

#include "obnsl_engine.h"

namespace obnsl_internal {

// Records `msg` as the calling thread's last error and returns `st`.
obnsl_status fail(obnsl_status st, const std::string& msg);

}  // namespace obnsl_internal
