// helpers of the functional MEX stand-in (tests/stub_mex/mex_fake.cpp)
#pragma once
#include <string>
#include <vector>
#include "mex.h"
mxArray* fake_matrix(size_t m, size_t n, const std::vector<double>& colmajor);
mxArray* fake_fill(size_t m, size_t n, double v);
mxArray* fake_string(const char* s);
mxArray* fake_struct();
mxArray* fake_field(const mxArray* a, const char* name);
extern std::vector<std::string> g_mex_warnings;
