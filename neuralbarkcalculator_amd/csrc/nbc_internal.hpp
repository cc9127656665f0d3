// Internal declarations shared between the translation units of libnbc_hip.so.
#pragma once
#include <cstdint>
#include <string>

namespace nbc {

extern thread_local std::string g_last_error;
int set_error(int code, const std::string& msg);
uint16_t f32_to_bf16(float f);

}  // namespace nbc
