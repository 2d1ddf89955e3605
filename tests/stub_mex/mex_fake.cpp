// Functional stand-in of the MEX C API declared in tests/stub_mex/mex.h: a small in-memory mxArray (double matrices, scalars, char
// rows, 1x1 structs).  Test infrastructure only (tests/stub_mex/run_gateways.cpp); mexErrMsgTxt throws, as MATLAB's long-jumps.
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>
#include "mex.h"
#include "mex_fake.h"

struct mxArray_tag {
  size_t m = 0, n = 0;
  bool is_char = false, is_struct = false;
  std::vector<double> data;
  std::string str;
  std::map<std::string, mxArray*> fields;
};
std::vector<std::string> g_mex_warnings;

mxArray* fake_matrix(size_t m, size_t n, const std::vector<double>& colmajor) { mxArray* a = new mxArray_tag; a->m = m; a->n = n; a->data = colmajor; a->data.resize(m * n); return a; }
mxArray* fake_fill(size_t m, size_t n, double v) { return fake_matrix(m, n, std::vector<double>(m * n, v)); }
mxArray* fake_string(const char* s) { mxArray* a = new mxArray_tag; a->is_char = true; a->str = s; a->m = 1; a->n = std::strlen(s); return a; }
mxArray* fake_struct() { mxArray* a = new mxArray_tag; a->is_struct = true; a->m = a->n = 1; return a; }
mxArray* fake_field(const mxArray* a, const char* name) { auto it = a->fields.find(name); return it == a->fields.end() ? nullptr : it->second; }

extern "C" {
void mexErrMsgTxt(const char* msg) { throw std::runtime_error(msg); }
void mexWarnMsgTxt(const char* msg) { g_mex_warnings.push_back(msg); }
size_t mxGetM(const mxArray* a) { return a->m; }
size_t mxGetN(const mxArray* a) { return a->n; }
double* mxGetPr(const mxArray* a) { return const_cast<double*>(a->data.data()); }
double mxGetScalar(const mxArray* a) { return a->data.empty() ? 0.0 : a->data[0]; }
mwIndex* mxGetIr(const mxArray*) { return nullptr; }
mwIndex* mxGetJc(const mxArray*) { return nullptr; }
bool mxIsSparse(const mxArray*) { return false; }
bool mxIsDouble(const mxArray* a) { return !a->is_char && !a->is_struct; }
bool mxIsComplex(const mxArray*) { return false; }
bool mxIsStruct(const mxArray* a) { return a->is_struct; }
bool mxIsEmpty(const mxArray* a) { return a->m * a->n == 0; }
bool mxIsChar(const mxArray* a) { return a->is_char; }
int mxGetString(const mxArray* a, char* buf, mwSize buflen) { if (!a->is_char || buflen == 0) return 1; std::strncpy(buf, a->str.c_str(), buflen - 1); buf[buflen - 1] = 0; return 0; }
mxArray* mxGetField(const mxArray* a, mwIndex, const char* fieldname) { return fake_field(a, fieldname); }
mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity) { return fake_fill(m, n, 0.0); }
mxArray* mxCreateDoubleScalar(double v) { return fake_fill(1, 1, v); }
mxArray* mxCreateStructMatrix(mwSize, mwSize, int nfields, const char** fieldnames) { mxArray* a = fake_struct(); for (int i = 0; i < nfields; ++i) a->fields[fieldnames[i]] = nullptr; return a; }
void mxSetField(mxArray* a, mwIndex, const char* fieldname, mxArray* value) { a->fields[fieldname] = value; }
}
