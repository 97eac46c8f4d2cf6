// third-party stand-in (compile-only check): the CBLAS declarations include/caffe/util/mkl_alternate.hpp and math_functions.hpp use
#pragma once
extern "C" {
typedef enum { CblasRowMajor = 101, CblasColMajor = 102 } CBLAS_ORDER;
typedef enum { CblasNoTrans = 111, CblasTrans = 112, CblasConjTrans = 113 } CBLAS_TRANSPOSE;
void cblas_sscal(int n, float a, float* x, int incx);
void cblas_dscal(int n, double a, double* x, int incx);
void cblas_saxpy(int n, float a, const float* x, int incx, float* y, int incy);
void cblas_daxpy(int n, double a, const double* x, int incx, double* y, int incy);
}
