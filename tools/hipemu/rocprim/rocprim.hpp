// tools/hipemu — TEST INFRASTRUCTURE ONLY.  Host stand-in for the one rocPRIM primitive libtfgpu calls
// (tf_collapse.hip: radix_sort_pairs, a stable key sort).
#pragma once
#include <algorithm>
#include <numeric>
#include <vector>
namespace rocprim {
template <class K, class V>
inline hipError_t radix_sort_pairs(void *tmp, size_t &tmp_bytes, const K *keys_in, K *keys_out, const V *vals_in, V *vals_out, size_t n,
                                   unsigned begin_bit, unsigned end_bit, hipStream_t) {
  if (!tmp) { tmp_bytes = 16; return hipSuccess; }
  const K mask = end_bit >= sizeof(K) * 8 ? ~K(0) : (K)((K(1) << end_bit) - 1);
  std::vector<size_t> idx(n);
  std::iota(idx.begin(), idx.end(), (size_t)0);
  std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return ((keys_in[a] & mask) >> begin_bit) < ((keys_in[b] & mask) >> begin_bit); });
  std::vector<K> k(n); std::vector<V> v(n);
  for (size_t i = 0; i < n; i++) { k[i] = keys_in[idx[i]]; v[i] = vals_in[idx[i]]; }
  std::copy(k.begin(), k.end(), keys_out); std::copy(v.begin(), v.end(), vals_out);
  return hipSuccess;
}
}  // namespace rocprim
