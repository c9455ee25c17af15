/* Declarations of the part of MATLAB's C MEX / Matrix API that mex/bds_mex.c uses.  Two uses in an image that has no MATLAB:
 *   tests/test_mex_syntax.py   -fsyntax-only compile of the gateway against these declarations and include/bds_mi355x.h;
 *   tests/test_mex_mock.py     the gateway compiled for real against tests/mex_stub/mex_mock.c -- a small stand-in for the MEX
 *                              runtime (arrays, structs, strings, error exit by longjmp) -- and EXECUTED through ctypes: what
 *                              MATLAB would hand to mexFunction goes in, what it would get back is compared with the ctypes host path.
 * Neither is MATLAB: semantics are those documented for the R2018a interleaved-complex API. */
#ifndef BDS_TEST_MEX_STUB_H
#define BDS_TEST_MEX_STUB_H
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef enum { mxREAL, mxCOMPLEX } mxComplexity;
typedef enum { mxDOUBLE_CLASS = 6, mxINT8_CLASS = 8, mxINT32_CLASS = 12 } mxClassID;
void mexErrMsgIdAndTxt(const char *id, const char *fmt, ...);
int mexAtExit(void (*fn)(void));
mxArray *mxGetField(const mxArray *s, size_t index, const char *name);
double mxGetScalar(const mxArray *a);
size_t mxGetNumberOfElements(const mxArray *a);
double *mxGetDoubles(const mxArray *a);
int8_t *mxGetInt8s(const mxArray *a);
int32_t *mxGetInt32s(const mxArray *a);
uint16_t *mxGetChars(const mxArray *a);
bool mxIsInt8(const mxArray *a);
bool mxIsDouble(const mxArray *a);
bool mxIsNumeric(const mxArray *a);
bool mxIsLogical(const mxArray *a);
bool mxIsChar(const mxArray *a);
bool mxIsStruct(const mxArray *a);
int mxGetString(const mxArray *a, char *buf, size_t buflen);
mxArray *mxCreateDoubleMatrix(size_t m, size_t n, mxComplexity c);
mxArray *mxCreateNumericMatrix(size_t m, size_t n, mxClassID cls, mxComplexity c);
mxArray *mxCreateStructMatrix(size_t m, size_t n, int nfields, const char **names);
int mxAddField(mxArray *s, const char *name);
void mxSetField(mxArray *s, size_t index, const char *name, mxArray *v);
void mxDestroyArray(mxArray *a);
void *mxCalloc(size_t n, size_t size);
void mxFree(void *p);
mxArray *mxCreateString(const char *str);
mxArray *mxCreateLogicalScalar(bool v);
#endif
