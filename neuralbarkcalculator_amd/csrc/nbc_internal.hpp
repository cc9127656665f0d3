// Internal declarations shared between the translation units of libnbc_hip.so.
#pragma once
#include <cstdint>
#include <string>

namespace nbc {

extern thread_local std::string g_last_error;
int set_error(int code, const std::string& msg);
uint16_t f32_to_bf16(float f);
uint16_t f32_to_f16(float f);                       // IEEE binary16, round to nearest even, subnormals kept (= the device's (_Float16)x)
float f16_to_f32(uint16_t h);
void split_f16x2(float x, uint16_t* h0, uint16_t* h1);   // h0 = f16(x), h1 = f16((x - h0) * 2^11)
int f16x2_row_exponent(const float* row, size_t n, bool* clamped = nullptr);   // k with max |w| * 2^k in [2^14, 2^15): the row
                                                                            // normalisation of f16x2 weights; *clamped: k hit [-66, 80]

}  // namespace nbc
