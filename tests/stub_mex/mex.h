/* Minimal stand-in for MATLAB's mex.h: the declarations mex/qpOASES.cpp and mex/qpOASES_sequence.cpp use, written from the
 * documented MEX C API signatures.  tests/test_abi_cpu.py::test_mex_gateways_compile type-checks the gateways against it
 * (g++ -fsyntax-only); tests/stub_mex/mex_fake.cpp implements these functions on a small in-memory mxArray so that
 * tests/stub_mex/run_gateways.cpp can RUN the gateways on the CPU against a recording stand-in of libfsaempc. */
#ifndef FSAEMPC_STUB_MEX_H
#define FSAEMPC_STUB_MEX_H
#include <cstddef>
typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef size_t mwIndex;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
extern "C" {
void mexErrMsgTxt(const char* msg);
void mexWarnMsgTxt(const char* msg);
size_t mxGetM(const mxArray* a);
size_t mxGetN(const mxArray* a);
double* mxGetPr(const mxArray* a);
double mxGetScalar(const mxArray* a);
mwIndex* mxGetIr(const mxArray* a);
mwIndex* mxGetJc(const mxArray* a);
bool mxIsSparse(const mxArray* a);
bool mxIsDouble(const mxArray* a);
bool mxIsComplex(const mxArray* a);
bool mxIsStruct(const mxArray* a);
bool mxIsEmpty(const mxArray* a);
bool mxIsChar(const mxArray* a);
int mxGetString(const mxArray* a, char* buf, mwSize buflen);
mxArray* mxGetField(const mxArray* a, mwIndex index, const char* fieldname);
mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag);
mxArray* mxCreateDoubleScalar(double v);
mxArray* mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char** fieldnames);
void mxSetField(mxArray* a, mwIndex index, const char* fieldname, mxArray* value);
void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]);
}
#endif
