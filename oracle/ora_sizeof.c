/*
 * oracle/ora_sizeof.c — util.DeepSizeof over ColumnValues, restated.  TEST INFRASTRUCTURE ONLY (see ora.h).
 *
 * The reference measures a ChangeItem by reflection over its boxed values (pkg/util/sizeof.go:7-93):
 *   - measurer.AsyncPush sets Size.Values = DeepSizeof(item.ColumnValues) (pkg/middlewares/synchronizer/measurer.go:38-42),
 *     which the Bufferer's byte trigger sums;
 *   - the s3 CSV reader sets Size.Read = DeepSizeof(vals) on every row it builds (reader_csv.go:336).
 * For a []interface{} (sizeofSlice, sizeof.go:56-75): 24 bytes of slice header, then per element 16 bytes of
 * interface header + DeepSizeof of the dynamic value — 0 for a nil interface (reflect.Invalid, :31-32), the type's
 * size for scalars (:10-11), 16 + len for strings (json.Number is a string kind, :12-13), 24 + len for []byte
 * (sizeofSlice's simple-element branch, :60-63), 24 for time.Time (a struct whose three word-sized fields are
 * unexported: SizeOfStruct adds each field type's size, :49-51).  Decoded JSON (`any`): a map is 8 + per entry
 * (16 + len(key)) + 16 + DeepSizeof(value) (sizeofMap, :77-93: the map header is one pointer; duplicate keys of the
 * text were merged by the decoder), a []interface{} recursively as above, numbers as json.Number (UseNumber) or as
 * float64 = 8 bytes (flags & 1).
 * Pinned by the formulas TestDeepSizeof states (pkg/util/sizeof_test.go:12-140) in tests/test_host_logic.py.
 */
#include <stdlib.h>
#include <string.h>
#include "ora.h"
#include "ora_json.h"

static uint64_t deep_json(const jnode *n, int float_numbers) {
  switch (n->type) {
    case JN_NULL: return 0;
    case JN_BOOL: return 1;
    case JN_NUM: return float_numbers ? 8 : 16 + (uint64_t)n->slen;
    case JN_STR: return 16 + (uint64_t)n->slen;
    case JN_ARR: {
      uint64_t s = 24;
      for (int i = 0; i < n->n; i++) s += 16 + deep_json(n->kids[i], float_numbers);
      return s;
    }
    case JN_OBJ: {
      uint64_t s = 8;
      for (int i = 0; i < n->n; i++) {
        int later = 0;  /* a later duplicate replaces this entry in the decoded map */
        for (int j = i + 1; j < n->n && !later; j++) later = strcmp(n->keys[i], n->keys[j]) == 0;
        if (later) continue;
        s += 16 + strlen(n->keys[i]) + 16 + deep_json(n->kids[i], float_numbers);
      }
      return s;
    }
  }
  return 0;
}

uint64_t ora_deepsizeof_value(const ora_value *v, int flags) {
  switch (v->kind) {
    case OV_NIL: return 0;
    case OV_I8: case OV_U8: case OV_BOOL: return 1;
    case OV_I16: case OV_U16: return 2;
    case OV_I32: case OV_U32: case OV_F32: return 4;
    case OV_I64: case OV_U64: case OV_F64: case OV_DURATION: return 8;
    case OV_STRING: case OV_JSONNUM: return 16 + (uint64_t)v->slen;
    case OV_BYTES: return 24 + (uint64_t)v->slen;
    case OV_TIME: return 24;
    case OV_JSON: {
      char *z = (char *)malloc(v->slen + 1);
      memcpy(z, v->s, v->slen); z[v->slen] = 0;
      jnode *n = jn_parse(z, NULL, 0);
      free(z);
      if (!n) return 0;
      uint64_t s = deep_json(n, flags & 1);
      jn_free(n);
      return s;
    }
  }
  return 0;
}

/* DeepSizeof(item.ColumnValues) for every item; returns the sum */
uint64_t ora_deepsizeof(const ora_batch *b, int flags, uint64_t *per_row) {
  uint64_t total = 0;
  for (int64_t r = 0; r < b->n; r++) {
    const ora_item *it = &b->items[r];
    uint64_t s = 24;
    for (int c = 0; c < it->nvalues; c++) s += 16 + ora_deepsizeof_value(&it->values[c], flags);
    if (per_row) per_row[r] = s;
    total += s;
  }
  return total;
}
