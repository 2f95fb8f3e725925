/*
 * oracle/ora_hash.c — SHA-256 (FIPS 180-4), HMAC (RFC 2104), CRC-32/IEEE and
 * FNV-1a-32, i.e. Go's crypto/sha256, crypto/hmac, hash/crc32.IEEETable and
 * hash/fnv.New32a as used by mask/hmac_hasher.go:29-33, pkg/util/crc32.go:5-7
 * and vendor_patched/.../kafka-go/balancer.go:143-181.
 * TEST INFRASTRUCTURE ONLY (see ora.h).
 */
#include <string.h>
#include "ora.h"

static const uint32_t K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

typedef struct { uint32_t h[8]; uint8_t buf[64]; size_t blen; uint64_t total; } sha_ctx;

static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

static void sha_block(uint32_t h[8], const uint8_t *p) {
  uint32_t w[64];
  for (int i = 0; i < 16; i++) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
  for (int i = 16; i < 64; i++) {
    uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
    uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
  for (int i = 0; i < 64; i++) {
    uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
    uint32_t ch = (e & f) ^ (~e & g);
    uint32_t t1 = hh + S1 + ch + K[i] + w[i];
    uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
    uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
    uint32_t t2 = S0 + mj;
    hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

static void sha_init(sha_ctx *c) {
  static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  memcpy(c->h, iv, sizeof iv); c->blen = 0; c->total = 0;
}
static void sha_update(sha_ctx *c, const void *data, size_t n) {
  const uint8_t *p = (const uint8_t *)data;
  c->total += n;
  while (n) {
    size_t k = 64 - c->blen; if (k > n) k = n;
    memcpy(c->buf + c->blen, p, k); c->blen += k; p += k; n -= k;
    if (c->blen == 64) { sha_block(c->h, c->buf); c->blen = 0; }
  }
}
static void sha_final(sha_ctx *c, uint8_t out[32]) {
  uint64_t bits = c->total * 8;
  uint8_t pad = 0x80;
  sha_update(c, &pad, 1);
  uint8_t z = 0;
  while (c->blen != 56) sha_update(c, &z, 1);
  uint8_t len[8];
  for (int i = 0; i < 8; i++) len[i] = (uint8_t)(bits >> (56 - 8 * i));
  sha_update(c, len, 8);
  for (int i = 0; i < 8; i++) { out[4 * i] = (uint8_t)(c->h[i] >> 24); out[4 * i + 1] = (uint8_t)(c->h[i] >> 16); out[4 * i + 2] = (uint8_t)(c->h[i] >> 8); out[4 * i + 3] = (uint8_t)c->h[i]; }
}

void ora_sha256(const void *data, size_t n, uint8_t out[32]) {
  sha_ctx c; sha_init(&c); sha_update(&c, data, n); sha_final(&c, out);
}

/* hmac.New(sha256.New, key) is built PER VALUE in the reference
 * (mask/hmac_hasher.go:30): key schedule + 4 compressions per short value. */
void ora_hmac_sha256(const void *key, size_t klen, const void *msg, size_t mlen, uint8_t out[32]) {
  uint8_t k0[64] = {0};
  if (klen > 64) ora_sha256(key, klen, k0); else memcpy(k0, key, klen);
  uint8_t ipad[64], opad[64];
  for (int i = 0; i < 64; i++) { ipad[i] = k0[i] ^ 0x36; opad[i] = k0[i] ^ 0x5c; }
  sha_ctx c; uint8_t inner[32];
  sha_init(&c); sha_update(&c, ipad, 64); sha_update(&c, msg, mlen); sha_final(&c, inner);
  sha_init(&c); sha_update(&c, opad, 64); sha_update(&c, inner, 32); sha_final(&c, out);
}

uint32_t ora_crc32_ieee(const void *data, size_t n) {
  static uint32_t tab[256]; static int init = 0;
  if (!init) {
    for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; tab[i] = c; }
    init = 1;
  }
  const uint8_t *p = (const uint8_t *)data;
  uint32_t c = 0xFFFFFFFFu;
  for (size_t i = 0; i < n; i++) c = tab[(c ^ p[i]) & 0xFF] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}

uint32_t ora_fnv1a32(const void *data, size_t n) {
  const uint8_t *p = (const uint8_t *)data;
  uint32_t h = 2166136261u;
  for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 16777619u; }
  return h;
}
