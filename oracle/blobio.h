/*
 * oracle/blobio.h -- tiny named-array container used for golden fixtures.
 * TEST INFRASTRUCTURE ONLY (written by oracle/ref_driver.cpp, read by
 * tests/golden_io.py).  Format, little endian:
 *   "SMGF1\n" then records { u32 name_len, name bytes, u8 dtype, u32 ndim,
 *   i64 dims[ndim], raw data }.  dtype: 'f' f32, 'd' f64, 'q' i64, 'i' i32,
 *   'B' u8, 'I' u32.
 */
#ifndef SMARTIES_AMD_ORACLE_BLOBIO_H
#define SMARTIES_AMD_ORACLE_BLOBIO_H
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include <stdexcept>

struct BlobWriter {
  FILE* f = nullptr;
  explicit BlobWriter(const std::string& path) {
    f = fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot open " + path);
    fwrite("SMGF1\n", 1, 6, f);
  }
  ~BlobWriter() { if (f) fclose(f); }
  static size_t esize(char dt) {
    switch (dt) { case 'f': case 'i': case 'I': return 4; case 'd': case 'q': return 8;
                  case 'B': return 1; }
    return 0;
  }
  void put(const std::string& name, char dt, const std::vector<int64_t>& dims, const void* data) {
    const uint32_t nl = (uint32_t)name.size(), nd = (uint32_t)dims.size();
    fwrite(&nl, 4, 1, f); fwrite(name.data(), 1, nl, f);
    fwrite(&dt, 1, 1, f); fwrite(&nd, 4, 1, f);
    size_t n = 1;
    for (auto d : dims) { fwrite(&d, 8, 1, f); n *= (size_t)d; }
    if (n) fwrite(data, esize(dt), n, f);
  }
  void f32(const std::string& n, const std::vector<float>& v, std::vector<int64_t> dims = {}) {
    if (dims.empty()) dims = {(int64_t)v.size()};
    put(n, 'f', dims, v.data());
  }
  void f64(const std::string& n, const std::vector<double>& v, std::vector<int64_t> dims = {}) {
    if (dims.empty()) dims = {(int64_t)v.size()};
    put(n, 'd', dims, v.data());
  }
  void i64(const std::string& n, const std::vector<int64_t>& v, std::vector<int64_t> dims = {}) {
    if (dims.empty()) dims = {(int64_t)v.size()};
    put(n, 'q', dims, v.data());
  }
  void u32(const std::string& n, const std::vector<uint32_t>& v, std::vector<int64_t> dims = {}) {
    if (dims.empty()) dims = {(int64_t)v.size()};
    put(n, 'I', dims, v.data());
  }
  void u8(const std::string& n, const std::vector<uint8_t>& v, std::vector<int64_t> dims = {}) {
    if (dims.empty()) dims = {(int64_t)v.size()};
    put(n, 'B', dims, v.data());
  }
  void scalar_d(const std::string& n, double x) { put(n, 'd', {1}, &x); }
  void scalar_q(const std::string& n, int64_t x) { put(n, 'q', {1}, &x); }
};
#endif
