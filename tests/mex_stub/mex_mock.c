/* A small stand-in for the MEX runtime (TEST INFRASTRUCTURE): the Matrix / MEX API functions mex/bds_mex.c calls, implemented
 * on plain C structs, so that the gateway can be compiled for real and EXECUTED in an image that has no MATLAB
 * (tests/test_mex_mock.py builds libbds_mex_mock.so = mex/bds_mex.c + this file, linked against the product library, and drives
 * mexFunction through ctypes).  Semantics follow the documented R2018a API for the calls used: column-major numeric arrays,
 * struct arrays with per-element fields, char arrays of 16-bit code units, mxGetString's 0 / 1 return, mexErrMsgIdAndTxt never
 * returning (here: longjmp back to mock_call, which reports identifier and message). */
#include <setjmp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mex.h"

enum { C_DOUBLE = 0, C_INT8, C_INT32, C_CHAR, C_STRUCT, C_LOGICAL };

struct mxArray_tag {
    int cls;
    size_t m, n;
    void *data;      /* numeric / char / logical payload */
    int nfields;     /* struct: field names and [element][field] values */
    char **names;
    mxArray **fields;
};

static jmp_buf g_jmp;
static int g_armed = 0;
static char g_err_id[128], g_err_msg[1024];
static void (*g_atexit)(void) = NULL;

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);

static size_t elsize(int cls) {
    switch (cls) {
        case C_DOUBLE: return 8;
        case C_INT32: return 4;
        case C_CHAR: return 2;
        default: return 1;
    }
}
static mxArray *make(int cls, size_t m, size_t n) {
    mxArray *a = (mxArray *)calloc(1, sizeof(mxArray));
    a->cls = cls, a->m = m, a->n = n;
    if (cls != C_STRUCT) a->data = calloc(m * n > 0 ? m * n : 1, elsize(cls));
    return a;
}

void mexErrMsgIdAndTxt(const char *id, const char *fmt, ...) {
    va_list ap;
    snprintf(g_err_id, sizeof(g_err_id), "%s", id ? id : "");
    va_start(ap, fmt);
    vsnprintf(g_err_msg, sizeof(g_err_msg), fmt, ap);
    va_end(ap);
    if (g_armed) longjmp(g_jmp, 1);
    fprintf(stderr, "mexErrMsgIdAndTxt outside mock_call: %s: %s\n", g_err_id, g_err_msg);
    abort();
}
int mexAtExit(void (*fn)(void)) {
    g_atexit = fn;
    return 0;
}

mxArray *mxGetField(const mxArray *s, size_t index, const char *name) {
    if (!s || s->cls != C_STRUCT || index >= s->m * s->n) return NULL;
    for (int k = 0; k < s->nfields; ++k)
        if (!strcmp(s->names[k], name)) return s->fields[index * (size_t)s->nfields + k];
    return NULL;
}
double mxGetScalar(const mxArray *a) {
    if (!a || a->m * a->n == 0) return 0.0; /* (MATLAB: undefined for empty input) */
    switch (a->cls) {
        case C_DOUBLE: return ((const double *)a->data)[0];
        case C_INT8: return ((const int8_t *)a->data)[0];
        case C_INT32: return ((const int32_t *)a->data)[0];
        case C_CHAR: return ((const uint16_t *)a->data)[0];
        case C_LOGICAL: return ((const uint8_t *)a->data)[0];
        default: return 0.0;
    }
}
size_t mxGetNumberOfElements(const mxArray *a) { return a ? a->m * a->n : 0; }
double *mxGetDoubles(const mxArray *a) { return a && a->cls == C_DOUBLE ? (double *)a->data : NULL; }
int8_t *mxGetInt8s(const mxArray *a) { return a && a->cls == C_INT8 ? (int8_t *)a->data : NULL; }
int32_t *mxGetInt32s(const mxArray *a) { return a && a->cls == C_INT32 ? (int32_t *)a->data : NULL; }
uint16_t *mxGetChars(const mxArray *a) { return a && a->cls == C_CHAR ? (uint16_t *)a->data : NULL; }
bool mxIsInt8(const mxArray *a) { return a && a->cls == C_INT8; }
bool mxIsDouble(const mxArray *a) { return a && a->cls == C_DOUBLE; }
bool mxIsNumeric(const mxArray *a) { return a && (a->cls == C_DOUBLE || a->cls == C_INT8 || a->cls == C_INT32); }
bool mxIsLogical(const mxArray *a) { return a && a->cls == C_LOGICAL; }
bool mxIsChar(const mxArray *a) { return a && a->cls == C_CHAR; }
bool mxIsStruct(const mxArray *a) { return a && a->cls == C_STRUCT; }
int mxGetString(const mxArray *a, char *buf, size_t buflen) {
    if (!a || a->cls != C_CHAR || buflen == 0) return 1;
    const size_t n = a->m * a->n;
    const uint16_t *c = (const uint16_t *)a->data;
    size_t i;
    for (i = 0; i < n && i + 1 < buflen; ++i) buf[i] = (char)c[i];
    buf[i] = 0;
    return n + 1 > buflen ? 1 : 0; /* 1: truncated */
}
mxArray *mxCreateDoubleMatrix(size_t m, size_t n, mxComplexity c) {
    (void)c;
    return make(C_DOUBLE, m, n);
}
mxArray *mxCreateNumericMatrix(size_t m, size_t n, mxClassID cls, mxComplexity c) {
    (void)c;
    return make(cls == mxINT8_CLASS ? C_INT8 : cls == mxINT32_CLASS ? C_INT32 : C_DOUBLE, m, n);
}
mxArray *mxCreateStructMatrix(size_t m, size_t n, int nfields, const char **names) {
    mxArray *a = make(C_STRUCT, m, n);
    for (int k = 0; k < nfields; ++k) mxAddField(a, names[k]);
    return a;
}
int mxAddField(mxArray *s, const char *name) {
    if (!s || s->cls != C_STRUCT) return -1;
    const size_t ne = s->m * s->n;
    const int nf = s->nfields + 1;
    mxArray **f = (mxArray **)calloc(ne * (size_t)nf > 0 ? ne * (size_t)nf : 1, sizeof(mxArray *));
    for (size_t e = 0; e < ne; ++e)
        for (int k = 0; k < s->nfields; ++k) f[e * (size_t)nf + k] = s->fields[e * (size_t)s->nfields + k];
    free(s->fields);
    s->fields = f;
    s->names = (char **)realloc(s->names, sizeof(char *) * (size_t)nf);
    s->names[nf - 1] = strdup(name);
    s->nfields = nf;
    return nf - 1;
}
void mxSetField(mxArray *s, size_t index, const char *name, mxArray *v) {
    if (!s || s->cls != C_STRUCT || index >= s->m * s->n) return;
    for (int k = 0; k < s->nfields; ++k)
        if (!strcmp(s->names[k], name)) s->fields[index * (size_t)s->nfields + k] = v;
}
void mxDestroyArray(mxArray *a) {
    if (!a) return;
    if (a->cls == C_STRUCT) {
        for (size_t i = 0; i < a->m * a->n * (size_t)a->nfields; ++i) mxDestroyArray(a->fields[i]);
        for (int k = 0; k < a->nfields; ++k) free(a->names[k]);
        free(a->names);
        free(a->fields);
    }
    free(a->data);
    free(a);
}
void *mxCalloc(size_t n, size_t size) { return calloc(n ? n : 1, size ? size : 1); }
void mxFree(void *p) { free(p); }
mxArray *mxCreateString(const char *str) {
    const size_t n = strlen(str);
    mxArray *a = make(C_CHAR, 1, n);
    for (size_t i = 0; i < n; ++i) ((uint16_t *)a->data)[i] = (unsigned char)str[i];
    return a;
}
mxArray *mxCreateLogicalScalar(bool v) {
    mxArray *a = make(C_LOGICAL, 1, 1);
    ((uint8_t *)a->data)[0] = v ? 1 : 0;
    return a;
}

/* ---- the test driver's side ---------------------------------------------------------------------------------------- */
/* mexFunction with MATLAB's error exit: 0 = returned normally, 1 = left through mexErrMsgIdAndTxt (identifier / message below) */
int mock_call(int nlhs, mxArray **plhs, int nrhs, const mxArray **prhs) {
    g_err_id[0] = g_err_msg[0] = 0;
    if (setjmp(g_jmp)) {
        g_armed = 0;
        return 1;
    }
    g_armed = 1;
    mexFunction(nlhs, plhs, nrhs, prhs);
    g_armed = 0;
    return 0;
}
const char *mock_error_id(void) { return g_err_id; }
const char *mock_error_msg(void) { return g_err_msg; }
void mock_run_atexit(void) { /* what MATLAB does when the MEX file is cleared */
    if (g_atexit) g_atexit();
    g_atexit = NULL;
}
size_t mock_rows(const mxArray *a) { return a ? a->m : 0; }
size_t mock_cols(const mxArray *a) { return a ? a->n : 0; }
int mock_nfields(const mxArray *a) { return a && a->cls == C_STRUCT ? a->nfields : -1; }
const char *mock_field_name(const mxArray *a, int k) { return a && a->cls == C_STRUCT && k >= 0 && k < a->nfields ? a->names[k] : NULL; }
