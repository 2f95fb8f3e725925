// tf_jsonscan.hpp — encoding/json on the device, shared by the JSON-envelope parsers (tf_srjson.hip, tf_debezium.hip):
// the scanner grammar (scanner.go: strings, numbers, literals, containers with an explicit stack), decoded strings as rune
// streams (decode.go unquote: escapes, surrogate pairs, invalid UTF-8 → U+FFFD), json.Marshal of a decoded value
// (appendString with escapeHTML; objects re-emitted with sorted keys where the source order differs) and
// strconv.ParseInt of a number literal.  Lane-per-item device functions over a MemBytes window.
#pragma once
#include "tf_common.hpp"
#include "tf_devparse.hpp"

namespace tf {
namespace sr {

enum : uint8_t { ST_OK = 0, ST_OTHER = 255, ST_DROPPED = 254 };  // else a tfgpu_rowerr
enum : uint32_t { VT_ABSENT = 0, VT_NULL, VT_FALSE, VT_TRUE, VT_NUM, VT_STR, VT_ARR, VT_OBJ,
                  VT_PLAIN = 0x80 /* flag on VT_STR: ASCII without escapes — the decoded string is the raw body */,
                  VT_CANON = 0x40 /* flag on VT_OBJ / VT_ARR under an `any` column: some object's keys are not in ascending
                                     order (or repeat), json.Marshal text needs the sorting emitter */, VT_MASK = 0x3F };
constexpr int CANON_DEPTH = 16;  // containers nested deeper than this under an unsorted `any` value go to the host
constexpr int MAX_DEPTH = 128;  // deeper containers (Go allows 10000) go to the host

static __device__ __forceinline__ bool is_ws(uint32_t c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n'; }
static __device__ __forceinline__ int hexv(uint32_t c) { return c >= '0' && c <= '9' ? (int)c - '0' : c >= 'a' && c <= 'f' ? (int)c - 'a' + 10 : c >= 'A' && c <= 'F' ? (int)c - 'A' + 10 : -1; }

// ---- encoding/json's grammar (scanner.go) ------------------------------------------------------------------------------
// String literal at pos (the opening quote): stateInString / stateInStringEsc*.  pos → past the closing quote.
static __device__ bool scan_string(MemBytes &rd, uint32_t &pos, const uint32_t end, bool *plain = nullptr) {
  const uint64_t ONES = 0x0101010101010101ull, HI = 0x8080808080808080ull;
  pos++;
  bool pl = true;  // no escapes, no bytes >= 0x80: the decoded string is the raw body
  for (;;) {
    // eight bytes per step while nothing in them needs a decision: not '"', not '\\', not a control byte (SWAR flags are
    // exact up to and including the first hit, so "no flag" is exact)
    while (pos + 8 <= end) {
      const uint64_t w = rd.word(pos);
      const uint64_t xq = w ^ (ONES * '"'), xb = w ^ (ONES * '\\');
      const uint64_t stop = (((xq - ONES) & ~xq) | ((xb - ONES) & ~xb) | ((w - ONES * 0x20) & ~w)) & HI;
      if (stop) { const uint32_t k = (uint32_t)(__ffsll((long long)stop) - 1) >> 3; if (k && (w & HI & ((1ull << (8 * k)) - 1))) pl = false; pos += k; break; }
      if (w & HI) pl = false;
      pos += 8;
    }
    if (pos >= end) return false;
    const uint32_t c = rd.at(pos);
    if (c == '"') { pos++; if (plain) *plain = pl; return true; }
    if (c < 0x20) return false;
    if (c >= 0x80) pl = false;
    if (c != '\\') { pos++; continue; }
    pl = false;
    if (pos + 1 >= end) return false;
    const uint32_t d = rd.at(pos + 1);
    pos += 2;
    if (d == 'u') {
      if (end - pos < 4) return false;
      for (int i = 0; i < 4; i++) if (hexv(rd.at(pos + i)) < 0) return false;
      pos += 4;
    } else if (!(d == '"' || d == '\\' || d == '/' || d == 'b' || d == 'f' || d == 'n' || d == 'r' || d == 't')) return false;
  }
}
// -?(0|[1-9][0-9]*)(\.[0-9]+)?([eE][+-]?[0-9]+)?
static __device__ bool scan_number(MemBytes &rd, uint32_t &pos, const uint32_t end) {
  auto dig = [&](uint32_t q) { if (q >= end) return false; const uint32_t c = rd.at(q); return c >= '0' && c <= '9'; };
  if (pos < end && rd.at(pos) == '-') pos++;
  if (pos >= end) return false;
  if (rd.at(pos) == '0') pos++;
  else if (dig(pos)) { while (dig(pos)) pos++; }
  else return false;
  if (pos < end && rd.at(pos) == '.') { pos++; if (!dig(pos)) return false; while (dig(pos)) pos++; }
  if (pos < end && (rd.at(pos) == 'e' || rd.at(pos) == 'E')) {
    pos++;
    if (pos < end && (rd.at(pos) == '+' || rd.at(pos) == '-')) pos++;
    if (!dig(pos)) return false;
    while (dig(pos)) pos++;
  }
  return true;
}
static __device__ bool scan_literal(MemBytes &rd, uint32_t &pos, const uint32_t end, uint32_t &vt) {
  const uint32_t c = rd.at(pos);
  const char *lit = c == 't' ? "true" : c == 'f' ? "false" : "null";
  const uint32_t n = c == 'f' ? 5u : 4u;
  if (end - pos < n) return false;
  for (uint32_t i = 0; i < n; i++) if (rd.at(pos + i) != (uint8_t)lit[i]) return false;
  pos += n;
  vt = c == 't' ? VT_TRUE : c == 'f' ? VT_FALSE : VT_NULL;
  return true;
}
// One value at pos (white space already skipped), containers walked with an explicit stack (bit = 1: object).
// 0 ok, 1 syntax error, 2 nesting deeper than MAX_DEPTH.
static __device__ int skip_value(MemBytes &rd, uint32_t &pos, const uint32_t end, uint32_t &vt) {
  uint64_t stk[MAX_DEPTH / 64] = {0, 0};
  int depth = 0;
  auto top_is_obj = [&]() { return (stk[(depth - 1) >> 6] >> ((depth - 1) & 63)) & 1; };
  auto skip_ws = [&]() { while (pos < end && is_ws(rd.at(pos))) pos++; };
  bool first = true;
  for (;;) {  // expect a value
    skip_ws();
    if (pos >= end) return 1;
    const uint32_t c = rd.at(pos);
    uint32_t t = 0;
    bool opened = false;
    if (c == '{' || c == '[') {
      if (depth == MAX_DEPTH) return 2;
      const uint64_t bit = 1ull << (depth & 63);
      if (c == '{') stk[depth >> 6] |= bit; else stk[depth >> 6] &= ~bit;
      depth++; pos++;
      t = c == '{' ? VT_OBJ : VT_ARR;
      skip_ws();
      if (pos >= end) return 1;
      if (rd.at(pos) == (c == '{' ? '}' : ']')) { pos++; depth--; }
      else opened = true;
    } else if (c == '"') { bool pl = false; if (!scan_string(rd, pos, end, &pl)) return 1; t = VT_STR | (pl ? VT_PLAIN : 0u); }
    else if (c == '-' || (c >= '0' && c <= '9')) { if (!scan_number(rd, pos, end)) return 1; t = VT_NUM; }
    else if (c == 't' || c == 'f' || c == 'n') { if (!scan_literal(rd, pos, end, t)) return 1; }
    else return 1;
    if (first) { vt = t; first = false; }
    if (!opened) {
      // after a value: close containers / move to the next element
      for (;;) {
        if (depth == 0) return 0;
        skip_ws();
        if (pos >= end) return 1;
        const uint32_t d = rd.at(pos);
        if (d == ',') { pos++; break; }
        if (d == (top_is_obj() ? '}' : ']')) { pos++; depth--; continue; }
        return 1;
      }
    }
    if (top_is_obj()) {  // a key, then ':'
      skip_ws();
      if (pos >= end || rd.at(pos) != '"') return 1;
      if (!scan_string(rd, pos, end)) return 1;
      skip_ws();
      if (pos >= end || rd.at(pos) != ':') return 1;
      pos++;
    }
  }
}

// ---- decoded strings (decode.go unquote) as rune streams ----------------------------------------------------------------
// Runes of a validated JSON string body [p, e) (escapes, surrogate pairs, invalid UTF-8 → U+FFFD), or of plain bytes.
struct RuneIter {
  MemBytes *rd; const uint8_t *plain; uint32_t p, e;
  __device__ __forceinline__ uint32_t at(uint32_t q) { return plain ? plain[q] : rd->at(q); }
  __device__ int next() {
    if (p >= e) return -1;
    const uint32_t c = at(p);
    if (!plain && c == '\\') {
      const uint32_t d = at(p + 1);
      p += 2;
      switch (d) {
        case 'b': return '\b'; case 'f': return '\f'; case 'n': return '\n'; case 'r': return '\r'; case 't': return '\t';
        case 'u': {
          int r = 0;
          for (int i = 0; i < 4; i++) r = r * 16 + hexv(at(p + i));
          p += 4;
          if (r >= 0xD800 && r < 0xE000) {
            if (r < 0xDC00 && e - p >= 6 && at(p) == '\\' && at(p + 1) == 'u') {
              int r2 = 0;
              for (int i = 0; i < 4; i++) r2 = r2 * 16 + hexv(at(p + 2 + i));
              if (r2 >= 0xDC00 && r2 < 0xE000) { p += 6; return 0x10000 + ((r - 0xD800) << 10) + (r2 - 0xDC00); }
            }
            return 0xFFFD;
          }
          return r;
        }
        default: return (int)d;  // " \ /
      }
    }
    if (c < 0x80) { p++; return (int)c; }
    uint32_t need = 0, cp = 0, lo = 0x80, hi = 0xBF;  // utf8.DecodeRune
    if (c >= 0xC2 && c <= 0xDF) { need = 1; cp = c & 0x1F; }
    else if (c >= 0xE0 && c <= 0xEF) { need = 2; cp = c & 0x0F; if (c == 0xE0) lo = 0xA0; if (c == 0xED) hi = 0x9F; }
    else if (c >= 0xF0 && c <= 0xF4) { need = 3; cp = c & 0x07; if (c == 0xF0) lo = 0x90; if (c == 0xF4) hi = 0x8F; }
    bool ok = need > 0 && e - p > need;
    if (ok) for (uint32_t k = 1; k <= need; k++) {
      const uint32_t d = at(p + k), l = k == 1 ? lo : 0x80u, h = k == 1 ? hi : 0xBFu;
      if (d < l || d > h) { ok = false; break; }
      cp = (cp << 6) | (d & 0x3F);
    }
    if (!ok) { p++; return 0xFFFD; }
    p += need + 1;
    return (int)cp;
  }
};
// <0, 0, >0: the decoded strings compared the way Go compares strings (UTF-8 bytes order = rune order)
static __device__ int rune_compare(RuneIter a, RuneIter b) {
  for (;;) {
    const int x = a.next(), y = b.next();
    if (x != y) return x < y ? -1 : 1;  // -1 = end sorts first
    if (x < 0) return 0;
  }
}
struct CountSink { uint32_t n = 0; __device__ __forceinline__ void put(uint32_t) { n++; } };
// Bytes leave eight at a time through one (possibly unaligned) 8-byte store, like tf_emit.hpp's WriteSink.  (A sink that
// bumps a uint8_t* per byte next to MemBytes' cached window was miscompiled for gfx950 at -O3 — the pointer was clobbered
// after the first store; a lane-at-a-time CPU run of the kernel cannot see that, the MI355X run did.)
struct ByteSink {
  uint8_t *p; uint64_t acc = 0; uint32_t n = 0;
  struct __attribute__((packed, aligned(1))) U64 { uint64_t v; };
  __device__ __forceinline__ void put(uint32_t c) {
    acc |= (uint64_t)(c & 0xFFu) << (8 * n);
    if (++n == 8) { reinterpret_cast<U64 *>(p)->v = acc; p += 8; acc = 0; n = 0; }
  }
  // the low k (1..8) bytes of w, the bytes above them zero
  __device__ __forceinline__ void put_word(uint64_t w, uint32_t k) {
    acc |= w << (8 * n);
    const uint32_t t = n + k;
    if (t >= 8) { reinterpret_cast<U64 *>(p)->v = acc; p += 8; acc = n ? w >> (8 * (8 - n)) : 0; n = t - 8; } else n = t;
  }
  __device__ __forceinline__ void flush() { for (; n; n--) { *p++ = (uint8_t)acc; acc >>= 8; } }
};
template <class S> __device__ __forceinline__ void put_utf8(S &o, uint32_t r) {
  if (r < 0x80) o.put(r);
  else if (r < 0x800) { o.put(0xC0 | (r >> 6)); o.put(0x80 | (r & 63)); }
  else if (r < 0x10000) { o.put(0xE0 | (r >> 12)); o.put(0x80 | ((r >> 6) & 63)); o.put(0x80 | (r & 63)); }
  else { o.put(0xF0 | (r >> 18)); o.put(0x80 | ((r >> 12) & 63)); o.put(0x80 | ((r >> 6) & 63)); o.put(0x80 | (r & 63)); }
}
template <class S> __device__ __forceinline__ void put_hex4(S &o, uint32_t r) {
  const char *H = "0123456789abcdef";
  o.put('\\'); o.put('u'); o.put(H[(r >> 12) & 15]); o.put(H[(r >> 8) & 15]); o.put(H[(r >> 4) & 15]); o.put(H[r & 15]);
}
// the unquoted string: Go string bytes
template <class S> __device__ void emit_unquoted(S &o, MemBytes &rd, uint32_t start, uint32_t len) {
  RuneIter it{&rd, nullptr, start + 1, start + len - 1};
  for (int r; (r = it.next()) >= 0;) put_utf8(o, (uint32_t)r);
}
// json.Marshal(string): encoding/json appendString with escapeHTML
template <class S> __device__ void emit_go_string(S &o, MemBytes &rd, uint32_t start, uint32_t len) {
  RuneIter it{&rd, nullptr, start + 1, start + len - 1};
  o.put('"');
  for (int r; (r = it.next()) >= 0;) {
    const uint32_t c = (uint32_t)r;
    if (c < 0x80) {
      if (c >= 0x20 && c != '"' && c != '\\' && c != '<' && c != '>' && c != '&') { o.put(c); continue; }
      switch (c) {
        case '"': o.put('\\'); o.put('"'); break; case '\\': o.put('\\'); o.put('\\'); break; case '\b': o.put('\\'); o.put('b'); break;
        case '\f': o.put('\\'); o.put('f'); break; case '\n': o.put('\\'); o.put('n'); break; case '\r': o.put('\\'); o.put('r'); break;
        case '\t': o.put('\\'); o.put('t'); break;
        default: put_hex4(o, c);
      }
    } else if (c == 0x2028 || c == 0x2029) put_hex4(o, c);
    else put_utf8(o, c);
  }
  o.put('"');
}
// json.Marshal of a decoded value whose objects already hold their keys in ascending order: the source tokens, strings
// re-encoded, white space dropped.  (The value was validated by skip_value.)
template <class S> __device__ void emit_any(S &o, MemBytes &rd, uint32_t start, uint32_t len) {
  uint32_t pos = start; const uint32_t end = start + len;
  while (pos < end) {
    const uint32_t c = rd.at(pos);
    if (is_ws(c)) { pos++; continue; }
    if (c == '"') { uint32_t q = pos; scan_string(rd, q, end); emit_go_string(o, rd, pos, q - pos); pos = q; continue; }
    o.put(c); pos++;
  }
}
// Can emit_any stand for json.Marshal?  Every object's keys must be strictly ascending (Go sorts map keys and a
// duplicate key keeps only its last value).  0: yes; 1: no — emit_any_canon sorts; 2: no, and nested deeper than
// emit_any_canon follows.  The value was validated by skip_value.
static __device__ int any_keys_order(MemBytes &rd, uint32_t start, uint32_t len) {
  uint32_t prev_s[MAX_DEPTH], prev_n[MAX_DEPTH];  // previous key of every open object (string bodies)
  uint64_t stk[MAX_DEPTH / 64] = {0, 0};
  int depth = 0, maxdepth = 0;
  bool sorted = true;
  uint32_t pos = start; const uint32_t end = start + len;
  bool expect_key = false;
  while (pos < end) {
    const uint32_t c = rd.at(pos);
    if (is_ws(c) || c == ':') { pos++; continue; }
    if (c == '{' || c == '[') {
      const uint64_t bit = 1ull << (depth & 63);
      if (c == '{') { stk[depth >> 6] |= bit; prev_n[depth] = 0xFFFFFFFFu; } else stk[depth >> 6] &= ~bit;
      depth++; pos++; expect_key = c == '{';
      if (depth > maxdepth) maxdepth = depth;
      continue;
    }
    if (c == '}' || c == ']') { depth--; pos++; expect_key = false; continue; }
    if (c == ',') { pos++; expect_key = depth > 0 && ((stk[(depth - 1) >> 6] >> ((depth - 1) & 63)) & 1); continue; }
    if (c == '"') {
      uint32_t q = pos;
      scan_string(rd, q, end);
      if (expect_key) {
        const int d = depth - 1;
        if (sorted && prev_n[d] != 0xFFFFFFFFu) {
          RuneIter a{&rd, nullptr, prev_s[d], prev_s[d] + prev_n[d]}, b{&rd, nullptr, pos + 1, q - 1};
          if (rune_compare(a, b) >= 0) sorted = false;
        }
        prev_s[d] = pos + 1; prev_n[d] = q - pos - 2;
        expect_key = false;
      }
      pos = q;
      continue;
    }
    pos++;  // number / literal bytes
  }
  return sorted ? 0 : maxdepth <= CANON_DEPTH ? 1 : 2;
}
// json.Marshal of a decoded value whose objects hold their keys in any order: what encoding/json does to a
// map[string]interface{} — keys ascending, the last duplicate's value — without materialising the map.  Every object is
// emitted by selection: the smallest key greater than the one emitted before, the last occurrence among equals
// (O(members^2) key compares, no storage per member).  Containers are followed with an explicit stack of CANON_DEPTH frames.
template <class S> __device__ void emit_any_canon(S &o, MemBytes &rd, uint32_t start, uint32_t len) {
  struct Frame { uint32_t s, e, cur, prev_n; uint8_t obj, first; };  // array: cur = next element; object: [cur, cur + prev_n) = last key emitted
  Frame st[CANON_DEPTH];
  int sp = 0;
  auto skip_ws = [&](uint32_t &q, uint32_t e) { while (q < e && is_ws(rd.at(q))) q++; };
  auto begin_value = [&](uint32_t vs, uint32_t ve) {
    const uint32_t c = rd.at(vs);
    if (c == '{' || c == '[') {
      if (sp == CANON_DEPTH) return;  // excluded by any_keys_order
      Frame &f = st[sp++];
      f.s = vs; f.e = ve; f.cur = vs + 1; f.prev_n = 0xFFFFFFFFu; f.obj = c == '{'; f.first = 1;
      o.put(c);
    } else if (c == '"') emit_go_string(o, rd, vs, ve - vs);
    else for (uint32_t q = vs; q < ve; q++) o.put(rd.at(q));  // number / literal token
  };
  begin_value(start, start + len);
  while (sp > 0) {
    Frame &f = st[sp - 1];
    if (!f.obj) {
      uint32_t pos = f.cur;
      skip_ws(pos, f.e);
      if (rd.at(pos) == ']') { o.put(']'); sp--; continue; }
      uint32_t q = pos, vt;
      skip_value(rd, q, f.e, vt);
      uint32_t nx = q;
      skip_ws(nx, f.e);
      if (rd.at(nx) == ',') nx++;
      f.cur = nx;
      if (!f.first) o.put(',');
      f.first = 0;
      begin_value(pos, q);
      continue;
    }
    // the next key of this object: the smallest one greater than the key emitted last, the last occurrence among equals
    uint32_t bks = 0, bkn = 0xFFFFFFFFu, bvs = 0, bve = 0;
    uint32_t pos = f.s + 1;
    for (;;) {
      skip_ws(pos, f.e);
      if (rd.at(pos) == '}') break;
      const uint32_t ks = pos;
      scan_string(rd, pos, f.e);
      const uint32_t ke = pos;
      skip_ws(pos, f.e);
      pos++;  // ':'
      skip_ws(pos, f.e);
      const uint32_t vs = pos;
      uint32_t vt;
      skip_value(rd, pos, f.e, vt);
      const uint32_t ve = pos;
      skip_ws(pos, f.e);
      if (rd.at(pos) == ',') pos++;
      bool after_prev = true;
      if (f.prev_n != 0xFFFFFFFFu) { RuneIter a{&rd, nullptr, f.cur, f.cur + f.prev_n}, b{&rd, nullptr, ks + 1, ke - 1}; after_prev = rune_compare(a, b) < 0; }
      if (!after_prev) continue;
      bool take = bkn == 0xFFFFFFFFu;
      if (!take) { RuneIter a{&rd, nullptr, ks + 1, ke - 1}, b{&rd, nullptr, bks, bks + bkn}; take = rune_compare(a, b) <= 0; }  // equal: the later one wins
      if (take) { bks = ks + 1; bkn = ke - ks - 2; bvs = vs; bve = ve; }
    }
    if (bkn == 0xFFFFFFFFu) { o.put('}'); sp--; continue; }
    if (!f.first) o.put(',');
    f.first = 0;
    f.cur = bks; f.prev_n = bkn;
    emit_go_string(o, rd, bks - 1, bkn + 2);
    o.put(':');
    begin_value(bvs, bve);
  }
}
// strconv.ParseInt(text, 10, 64) of a validated JSON number literal: false on a syntax or range error
static __device__ bool number_int64(MemBytes &rd, uint32_t start, uint32_t len, int64_t *out) {
  uint32_t i = 0; bool neg = false;
  if (rd.at(start) == '-') { neg = true; i = 1; }
  uint64_t v = 0;
  if (i >= len) return false;
  for (; i < len; i++) {
    const uint32_t c = rd.at(start + i);
    if (c < '0' || c > '9') return false;
    if (v > 1844674407370955161ull) return false;
    v = v * 10; const uint64_t d = c - '0';
    if (v > ~0ull - d) return false;
    v += d;
  }
  if (neg) { if (v > 9223372036854775808ull) return false; *out = (int64_t)(0 - v); }
  else { if (v > 9223372036854775807ull) return false; *out = (int64_t)v; }
  return true;
}


}  // namespace sr
}  // namespace tf
