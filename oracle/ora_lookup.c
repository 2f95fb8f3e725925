/*
 * oracle/ora_lookup.c — lookupComplex (pkg/parsers/generic/lookup.go:10-59): a nested ColSchema.Path ("EventValue.LogInfo",
 * "EventValue/PlaceCoordinates/lat") walked through a top-level STRING value that holds JSON — parseJSON's json.Unmarshal
 * into map[string]interface{}, retried after its two textual un-escapings.  TEST INFRASTRUCTURE ONLY (see ora.h): the checker
 * of the device's store_nested (tf_json.hip), pinned by the generic-parser canon cases `metrika` and `metrika_complex`.
 * Restated: targets that end at a JSON string or null, and every failure.  A number / bool / map / slice at the end is
 * reported as ORA_LOOKUP_OTHER (ParseVal of encoding/json's float64 and friends: not restated).
 */
#include "ora.h"
#include "ora_jv.h"

/* json.Unmarshal(data, &map[string]interface{}): the whole input is one value (checkValid), an object fills the map, `null`
 * leaves it nil (*is_null), anything else is an UnmarshalTypeError */
static jv *unmarshal_map(const char *p, size_t n, int *is_null) {
  jp s = {(const unsigned char *)p, (const unsigned char *)p + n, 0, 0};
  *is_null = 0;
  ws(&s);
  jv *v = parse_value(&s);
  if (!s.err) { ws(&s); if (s.p != s.e) s.err = 1; }  /* invalid character after top-level value */
  if (s.err || !v) { jv_free(v); return NULL; }
  if (v->t == JV_OBJ) return v;
  if (v->t == JV_NULL) *is_null = 1;
  jv_free(v);
  return NULL;
}
static char *replace_all(const char *s, size_t n, const char *from, const char *to, size_t *outn) {
  const size_t fl = strlen(from), tl = strlen(to);
  sbuf b = {0};
  sb_put(&b, "", 0);
  for (size_t i = 0; i < n;) {
    if (i + fl <= n && !memcmp(s + i, from, fl)) { sb_put(&b, to, tl); i += fl; }
    else { sb_c(&b, s[i]); i++; }
  }
  *outn = b.n;
  return b.p;
}
/* lookup.go:41-59 */
static jv *parse_json(const char *s, size_t n, int *is_null, int *err) {
  *err = 0;
  jv *m = unmarshal_map(s, n, is_null);
  if (m || *is_null) return m;
  size_t n2; char *s2 = replace_all(s, n, "\\\\\"", "\\\"", &n2);  /* \\" -> \" ("possible double escape") */
  m = unmarshal_map(s2, n2, is_null);
  if (m || *is_null) { free(s2); return m; }
  size_t n3; char *s3 = replace_all(s2, n2, "\\", "", &n3);
  free(s2);
  m = unmarshal_map(s3, n3, is_null);
  free(s3);
  if (!m && !*is_null) *err = 1;
  return m;
}

int ora_lookup_complex(const char *top, size_t topn, const char *const *segs, int nsegs, char **out, size_t *outn) {
  /* obj = item[segs[0]] was a Go string (top); the remaining field names walk it */
  jv *root = NULL;            /* owner of the tree the walk is in */
  const jv *obj = NULL;       /* NULL: the current value is the string (cur, curn) */
  char *cur = (char *)malloc(topn + 1); size_t curn = topn;
  memcpy(cur, top, topn); cur[topn] = 0;
  int rc = -1;
  for (int i = 1; i < nsegs && rc < 0; i++) {
    const jv *m = NULL;
    if (!obj) {
      int is_null, err;
      jv *parsed = parse_json(cur, curn, &is_null, &err);
      if (err) { rc = ORA_LOOKUP_ERROR; break; }   /* unable to parse json */
      if (!parsed) { rc = ORA_LOOKUP_ERROR; break; }  /* a nil map: m[fieldName] is not there — unable to get field */
      jv_free(root); root = parsed; m = parsed;
    } else if (obj->t == JV_OBJ) m = obj;
    else { rc = ORA_LOOKUP_ERROR; break; }        /* unexpected value type */
    const jv *next = NULL;
    const size_t sl = strlen(segs[i]);
    for (int k = 0; k < m->nk; k++) if (m->klen[k] == sl && !memcmp(m->keys[k], segs[i], sl)) next = m->kids[k];
    if (!next) { rc = ORA_LOOKUP_ERROR; break; }   /* unable to get field */
    if (next->t == JV_STR) { free(cur); curn = next->n; cur = (char *)malloc(curn + 1); memcpy(cur, next->s, curn); cur[curn] = 0; obj = NULL; }
    else obj = next;
  }
  if (rc < 0) {
    if (!obj) { *out = cur; *outn = curn; cur = NULL; rc = ORA_LOOKUP_STRING; }
    else rc = obj->t == JV_NULL ? ORA_LOOKUP_NIL : ORA_LOOKUP_OTHER;
  }
  free(cur); jv_free(root);
  return rc;
}
