/*
 * oracle/ora_value.c — the reference's row-oriented, boxed data model
 * (pkg/abstract/changeitem/change_item.go:27-80: ColumnValues []interface{};
 * table_schema.go:10-15; col_schema.go:14-29) restated in C, plus the
 * columnar<->row converters the tests use for I/O.
 * TEST INFRASTRUCTURE ONLY (see ora.h).
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ora.h"

static char *dupn(const char *s, size_t n) {
  char *r = (char *)malloc(n + 1);
  if (n) memcpy(r, s, n);
  r[n] = 0;
  return r;
}
static char *dups(const char *s) { return s ? dupn(s, strlen(s)) : NULL; }

void ora_value_free(ora_value *v) {
  if (v->s) free(v->s);
  v->s = NULL; v->kind = OV_NIL;
}
ora_value ora_value_clone(const ora_value *v) {
  ora_value r = *v;
  if (v->s) r.s = dupn(v->s, v->slen);
  return r;
}

ora_batch *ora_batch_new(void) { return (ora_batch *)calloc(1, sizeof(ora_batch)); }

/* a tiny string pool for the schema strings that are copied around by value */
const char *ora_intern(const char *s) {
  static char **pool = NULL; static int n = 0, cap = 0;
  if (!s) s = "";
  for (int i = 0; i < n; i++) if (strcmp(pool[i], s) == 0) return pool[i];
  if (n == cap) { cap = cap ? cap * 2 : 32; pool = (char **)realloc(pool, sizeof(char *) * (size_t)cap); }
  pool[n] = dups(s);
  return pool[n++];
}

void ora_schema_unref(ora_schema *s) {
  if (!s) return;
  if (--s->refs > 0) return;
  for (int i = 0; i < s->ncols; i++) { free(s->cols[i].name); free(s->cols[i].path); free(s->cols[i].original_type); }
  free(s->cols); free(s);
}
static void names_unref(ora_names *n) {
  if (!n) return;
  if (--n->refs > 0) return;
  for (int i = 0; i < n->n; i++) free(n->names[i]);
  free(n->names); free(n);
}

void ora_item_clear(ora_item *it) {
  for (int i = 0; i < it->nvalues; i++) ora_value_free(&it->values[i]);
  free(it->values);
  free(it->ns); free(it->table); free(it->part_id);
  names_unref(it->names);
  ora_schema_unref(it->schema);
  for (int i = 0; i < it->n_old; i++) ora_value_free(&it->old_values[i]);
  free(it->old_values);
  names_unref(it->old_names);
  memset(it, 0, sizeof *it);
}

void ora_batch_free(ora_batch *b) {
  if (!b) return;
  for (int64_t i = 0; i < b->n; i++) ora_item_clear(&b->items[i]);
  free(b->items);
  for (int64_t i = 0; i < b->nerr; i++) free(b->errs[i].msg);
  free(b->errs);
  free(b);
}

ora_item *ora_batch_push(ora_batch *b) {
  if (b->n == b->cap) {
    b->cap = b->cap ? b->cap * 2 : 16; /* Go append growth */
    b->items = (ora_item *)realloc(b->items, (size_t)b->cap * sizeof(ora_item));
  }
  ora_item *it = &b->items[b->n++];
  memset(it, 0, sizeof *it);
  return it;
}

void ora_batch_add_error(ora_batch *b, int64_t row, int code, const char *msg) {
  if (b->nerr == b->errcap) {
    b->errcap = b->errcap ? b->errcap * 2 : 16;
    b->errs = (ora_error *)realloc(b->errs, (size_t)b->errcap * sizeof(ora_error));
  }
  b->errs[b->nerr].row = row; b->errs[b->nerr].code = code; b->errs[b->nerr].msg = dups(msg);
  b->nerr++;
}

ora_schema *ora_schema_from(const tfgpu_schema *s) {
  ora_schema *r = (ora_schema *)calloc(1, sizeof *r);
  r->refs = 1;
  r->ncols = s ? s->ncols : 0;
  r->cols = (ora_colschema *)calloc((size_t)(r->ncols ? r->ncols : 1), sizeof(ora_colschema));
  for (int i = 0; i < r->ncols; i++) {
    r->cols[i].name = dups(s->cols[i].name);
    r->cols[i].dtype = s->cols[i].dtype;
    r->cols[i].key = (s->cols[i].flags & TFGPU_COL_KEY) != 0;
    r->cols[i].path = dups(s->cols[i].path ? s->cols[i].path : "");
    r->cols[i].original_type = dups(s->cols[i].original_type ? s->cols[i].original_type : "");
    r->cols[i].table_schema = ora_intern(s->cols[i].table_schema); r->cols[i].table_name = ora_intern(s->cols[i].table_name);
    r->cols[i].expression = ora_intern(s->cols[i].expression);
    r->cols[i].properties_json = s->cols[i].properties_json ? ora_intern(s->cols[i].properties_json) : NULL;
    r->cols[i].fake_key = (s->cols[i].flags & TFGPU_COL_FAKE_KEY) != 0; r->cols[i].required = (s->cols[i].flags & TFGPU_COL_REQUIRED) != 0;
  }
  return r;
}

static int is_var(int repr) { return repr == TFGPU_R_STRING || repr == TFGPU_R_BYTES || repr == TFGPU_R_JSONNUM || repr == TFGPU_R_JSON; }

static int valid_at(const uint8_t *bm, int64_t i) { return !bm || ((bm[i >> 3] >> (i & 7)) & 1); }

static ora_value box(const tfgpu_column *c, int64_t r) {
  ora_value v; memset(&v, 0, sizeof v);
  if (!valid_at(c->validity, r)) { v.kind = OV_NIL; return v; }
  switch (c->repr) {
    case TFGPU_R_INT8: v.kind = OV_I8; v.v.i = ((int8_t *)c->values)[r]; break;
    case TFGPU_R_INT16: v.kind = OV_I16; v.v.i = ((int16_t *)c->values)[r]; break;
    case TFGPU_R_INT32: v.kind = OV_I32; v.v.i = ((int32_t *)c->values)[r]; break;
    case TFGPU_R_INT64: v.kind = OV_I64; v.v.i = ((int64_t *)c->values)[r]; break;
    case TFGPU_R_UINT8: v.kind = OV_U8; v.v.u = ((uint8_t *)c->values)[r]; break;
    case TFGPU_R_UINT16: v.kind = OV_U16; v.v.u = ((uint16_t *)c->values)[r]; break;
    case TFGPU_R_UINT32: v.kind = OV_U32; v.v.u = ((uint32_t *)c->values)[r]; break;
    case TFGPU_R_UINT64: v.kind = OV_U64; v.v.u = ((uint64_t *)c->values)[r]; break;
    case TFGPU_R_FLOAT32: v.kind = OV_F32; v.v.f32 = ((float *)c->values)[r]; break;
    case TFGPU_R_FLOAT64: v.kind = OV_F64; v.v.f64 = ((double *)c->values)[r]; break;
    case TFGPU_R_BOOL: v.kind = OV_BOOL; v.v.b = ((uint8_t *)c->values)[r] != 0; break;
    case TFGPU_R_TIME: v.kind = OV_TIME; v.v.t.sec = ((int64_t *)c->values)[r]; v.v.t.nsec = c->nanos ? c->nanos[r] : 0; break;
    case TFGPU_R_DURATION: v.kind = OV_DURATION; v.v.i = ((int64_t *)c->values)[r]; break;
    case TFGPU_R_STRING: case TFGPU_R_BYTES: case TFGPU_R_JSONNUM: case TFGPU_R_JSON: {
      v.kind = c->repr == TFGPU_R_STRING ? OV_STRING : c->repr == TFGPU_R_BYTES ? OV_BYTES : c->repr == TFGPU_R_JSONNUM ? OV_JSONNUM : OV_JSON;
      uint32_t a = c->offsets[r], b = c->offsets[r + 1];
      v.slen = b - a; v.s = dupn((const char *)c->data + a, v.slen);
      break;
    }
    default: v.kind = OV_NIL;
  }
  return v;
}

ora_batch *ora_from_columns(const tfgpu_batch *cb, const tfgpu_schema *schema) {
  ora_batch *b = ora_batch_new();
  ora_schema *sch = ora_schema_from(schema);
  ora_names *nm = (ora_names *)calloc(1, sizeof *nm);
  nm->refs = 1; nm->n = cb->ncols; nm->names = (char **)calloc((size_t)(cb->ncols ? cb->ncols : 1), sizeof(char *));
  for (int i = 0; i < cb->ncols; i++) nm->names[i] = dups(cb->cols[i].name);
  ora_names *onm = NULL;
  if (cb->n_old_keys > 0) {
    onm = (ora_names *)calloc(1, sizeof *onm);
    onm->refs = 1; onm->n = cb->n_old_keys; onm->names = (char **)calloc((size_t)cb->n_old_keys, sizeof(char *));
    for (int i = 0; i < cb->n_old_keys; i++) onm->names[i] = dups(cb->old_keys[i].name);
  }
  b->cap = cb->nrows ? cb->nrows : 1;
  b->items = (ora_item *)calloc((size_t)b->cap, sizeof(ora_item));
  for (int64_t r = 0; r < cb->nrows; r++) {
    ora_item *it = ora_batch_push(b);
    it->kind = cb->kind ? cb->kind[r] : TFGPU_K_INSERT;
    it->ns = dups(cb->table_ns ? cb->table_ns : "");
    it->table = dups(cb->table_name ? cb->table_name : "");
    if (cb->part_id) { char pb[16]; snprintf(pb, sizeof pb, "%u", cb->part_id[r]); it->part_id = dups(pb); }  /* PartID = itoa (sharder.go:145) */
    else it->part_id = dups("");
    it->schema = sch; sch->refs++;
    int listed = 0;  /* tfgpu_column.absent: the row's ColumnNames are the columns it is not absent from, in batch order */
    for (int c = 0; c < cb->ncols; c++) if (!(cb->cols[c].absent && valid_at(cb->cols[c].absent, r))) listed++;
    if (listed == cb->ncols && !cb->col_order) {
      it->names = nm; nm->refs++;
      it->nvalues = cb->ncols;
      it->values = (ora_value *)calloc((size_t)(cb->ncols ? cb->ncols : 1), sizeof(ora_value));
      for (int c = 0; c < cb->ncols; c++) it->values[c] = box(&cb->cols[c], r);
    } else {
      ora_names *own = (ora_names *)calloc(1, sizeof *own);
      own->refs = 1; own->n = listed; own->names = (char **)calloc((size_t)(listed ? listed : 1), sizeof(char *));
      it->names = own;
      it->nvalues = listed;
      it->values = (ora_value *)calloc((size_t)(listed ? listed : 1), sizeof(ora_value));
      int k = 0;
      for (int q = 0; q < cb->ncols; q++) {
        const int c = cb->col_order ? cb->col_order[r * cb->ncols + q] : q;  /* tfgpu_batch.col_order: the row's own ColumnNames order */
        if (cb->cols[c].absent && valid_at(cb->cols[c].absent, r)) continue;
        own->names[k] = dups(cb->cols[c].name);
        it->values[k++] = box(&cb->cols[c], r);
      }
    }
    it->src_row = cb->src_row ? cb->src_row[r] : r;
    if (onm && valid_at(cb->old_keys_present, r)) {
      it->old_names = onm; onm->refs++;
      it->n_old = cb->n_old_keys;
      it->old_values = (ora_value *)calloc((size_t)cb->n_old_keys, sizeof(ora_value));
      for (int c = 0; c < cb->n_old_keys; c++) it->old_values[c] = box(&cb->old_keys[c], r);
    }
  }
  names_unref(onm);
  names_unref(nm);
  ora_schema_unref(sch);
  return b;
}

static int repr_of(int k) {
  switch (k) {
    case OV_I8: return TFGPU_R_INT8; case OV_I16: return TFGPU_R_INT16; case OV_I32: return TFGPU_R_INT32; case OV_I64: return TFGPU_R_INT64;
    case OV_U8: return TFGPU_R_UINT8; case OV_U16: return TFGPU_R_UINT16; case OV_U32: return TFGPU_R_UINT32; case OV_U64: return TFGPU_R_UINT64;
    case OV_F32: return TFGPU_R_FLOAT32; case OV_F64: return TFGPU_R_FLOAT64; case OV_BOOL: return TFGPU_R_BOOL;
    case OV_STRING: return TFGPU_R_STRING; case OV_BYTES: return TFGPU_R_BYTES; case OV_JSONNUM: return TFGPU_R_JSONNUM; case OV_JSON: return TFGPU_R_JSON;
    case OV_TIME: return TFGPU_R_TIME; case OV_DURATION: return TFGPU_R_DURATION;
  }
  return TFGPU_R_INVALID;
}
static size_t repr_width(int r) {
  switch (r) {
    case TFGPU_R_INT8: case TFGPU_R_UINT8: case TFGPU_R_BOOL: return 1;
    case TFGPU_R_INT16: case TFGPU_R_UINT16: return 2;
    case TFGPU_R_INT32: case TFGPU_R_UINT32: case TFGPU_R_FLOAT32: return 4;
    case TFGPU_R_INT64: case TFGPU_R_UINT64: case TFGPU_R_FLOAT64: case TFGPU_R_TIME: case TFGPU_R_DURATION: return 8;
  }
  return 0;
}

/* idx (ColumnValues only): where item r holds this column, -1 = the item does not list it (an ABSENT cell: nil here, its bit in col->absent) */
#define VAL(it) (old ? ((it)->n_old > c ? &(it)->old_values[c] : &NILV) : idx ? (idx[(it) - b->items] >= 0 ? &(it)->values[idx[(it) - b->items]] : &NILV) : &(it)->values[c])
static const ora_value NILV = {0};
static void fill_column(tfgpu_column *col, const ora_batch *b, const ora_item *f, const char *name, int c, int old, const int *idx) {
  {
    col->name = dups(name);
    /* dtype by name from the row's schema */
    col->dtype = TFGPU_T_INVALID;
    if (f->schema) for (int k = 0; k < f->schema->ncols; k++) if (!strcmp(f->schema->cols[k].name, col->name)) { col->dtype = f->schema->cols[k].dtype; break; }
    int repr = TFGPU_R_INVALID;
    for (int64_t r = 0; r < b->n && repr == TFGPU_R_INVALID; r++) repr = repr_of((*VAL(&b->items[r])).kind);
    if (repr == TFGPU_R_INVALID) repr = TFGPU_R_STRING; /* all-nil column */
    col->repr = repr;
    col->validity = (uint8_t *)calloc((size_t)((b->n + 7) / 8), 1);
    if (is_var(repr)) {
      col->offsets = (uint32_t *)calloc((size_t)b->n + 1, sizeof(uint32_t));
      uint64_t tot = 0;
      for (int64_t r = 0; r < b->n; r++) { const ora_value *v = &(*VAL(&b->items[r])); col->offsets[r] = (uint32_t)tot; if (v->kind != OV_NIL) tot += v->slen; }
      col->offsets[b->n] = (uint32_t)tot; col->data_len = tot;
      col->data = (uint8_t *)malloc(tot ? tot : 1);
      for (int64_t r = 0; r < b->n; r++) {
        const ora_value *v = &(*VAL(&b->items[r]));
        if (v->kind == OV_NIL) continue;
        col->validity[r >> 3] |= (uint8_t)(1u << (r & 7));
        if (v->slen) memcpy(col->data + col->offsets[r], v->s, v->slen);
      }
    } else {
      size_t w = repr_width(repr);
      col->values = calloc((size_t)b->n, w);
      if (repr == TFGPU_R_TIME) col->nanos = (int32_t *)calloc((size_t)b->n, sizeof(int32_t));
      for (int64_t r = 0; r < b->n; r++) {
        const ora_value *v = &(*VAL(&b->items[r]));
        if (v->kind == OV_NIL) continue;
        col->validity[r >> 3] |= (uint8_t)(1u << (r & 7));
        switch (repr) {
          case TFGPU_R_INT8: ((int8_t *)col->values)[r] = (int8_t)v->v.i; break;
          case TFGPU_R_INT16: ((int16_t *)col->values)[r] = (int16_t)v->v.i; break;
          case TFGPU_R_INT32: ((int32_t *)col->values)[r] = (int32_t)v->v.i; break;
          case TFGPU_R_INT64: case TFGPU_R_DURATION: ((int64_t *)col->values)[r] = v->v.i; break;
          case TFGPU_R_UINT8: ((uint8_t *)col->values)[r] = (uint8_t)v->v.u; break;
          case TFGPU_R_UINT16: ((uint16_t *)col->values)[r] = (uint16_t)v->v.u; break;
          case TFGPU_R_UINT32: ((uint32_t *)col->values)[r] = (uint32_t)v->v.u; break;
          case TFGPU_R_UINT64: ((uint64_t *)col->values)[r] = v->v.u; break;
          case TFGPU_R_FLOAT32: ((float *)col->values)[r] = v->v.f32; break;
          case TFGPU_R_FLOAT64: ((double *)col->values)[r] = v->v.f64; break;
          case TFGPU_R_BOOL: ((uint8_t *)col->values)[r] = (uint8_t)v->v.b; break;
          case TFGPU_R_TIME: ((int64_t *)col->values)[r] = v->v.t.sec; col->nanos[r] = v->v.t.nsec; break;
        }
      }
    }
  }
}
#undef VAL

tfgpu_batch *ora_to_columns(const ora_batch *b) {
  tfgpu_batch *cb = (tfgpu_batch *)calloc(1, sizeof *cb);
  cb->nrows = b->n; cb->mem = TFGPU_MEM_HOST;
  cb->kind = (uint8_t *)calloc((size_t)(b->n ? b->n : 1), 1);
  cb->src_row = (int32_t *)calloc((size_t)(b->n ? b->n : 1), sizeof(int32_t));
  cb->part_id = (uint32_t *)calloc((size_t)(b->n ? b->n : 1), sizeof(uint32_t));
  if (b->n == 0) return cb;
  const ora_item *f = &b->items[0];
  cb->table_ns = dups(f->ns); cb->table_name = dups(f->table);
  cb->ncols = f->nvalues;
  cb->cols = (tfgpu_column *)calloc((size_t)(cb->ncols ? cb->ncols : 1), sizeof(tfgpu_column));
  for (int64_t r = 0; r < b->n; r++) {
    cb->kind[r] = (uint8_t)b->items[r].kind;
    cb->src_row[r] = (int32_t)b->items[r].src_row;
    const char *p = b->items[r].part_id;
    cb->part_id[r] = (p && *p >= '0' && *p <= '9') ? (uint32_t)strtoul(p, NULL, 10) : 0;
  }
  /* items that list different columns (TOAST-style Updates, isGenerateUpdates, isAbsent): the batch's columns are the first item's TableSchema
   * (when every listed name is in it) or the names in order of first appearance; an item that does not list a column is ABSENT there */
  int ragged = 0;
  for (int64_t r = 1; r < b->n && !ragged; r++) {
    const ora_item *it = &b->items[r];
    if (it->names == f->names) continue;
    if (it->nvalues != f->nvalues) { ragged = 1; break; }
    for (int c = 0; c < it->nvalues; c++) if (strcmp(it->names->names[c], f->names->names[c])) { ragged = 1; break; }
  }
  if (ragged) {
    char **un = NULL; int nu = 0, capu = 0;
    int covered = f->schema != NULL;
    for (int64_t r = 0; r < b->n && covered; r++) for (int c = 0; c < b->items[r].nvalues && covered; c++) {
      int hit = 0;
      for (int k = 0; k < f->schema->ncols; k++) if (!strcmp(f->schema->cols[k].name, b->items[r].names->names[c])) { hit = 1; break; }
      if (!hit) covered = 0;
    }
    if (covered) { nu = f->schema->ncols; un = (char **)calloc((size_t)(nu ? nu : 1), sizeof(char *)); for (int k = 0; k < nu; k++) un[k] = f->schema->cols[k].name; }
    else for (int64_t r = 0; r < b->n; r++) for (int c = 0; c < b->items[r].nvalues; c++) {
      const char *nm = b->items[r].names->names[c];
      int hit = 0;
      for (int k = 0; k < nu; k++) if (!strcmp(un[k], nm)) { hit = 1; break; }
      if (hit) continue;
      if (nu == capu) { capu = capu ? capu * 2 : 16; un = (char **)realloc(un, sizeof(char *) * (size_t)capu); }
      un[nu++] = (char *)nm;
    }
    free(cb->cols);
    cb->ncols = nu;
    cb->cols = (tfgpu_column *)calloc((size_t)(nu ? nu : 1), sizeof(tfgpu_column));
    int *idx = (int *)malloc(sizeof(int) * (size_t)b->n);
    for (int k = 0; k < nu; k++) {
      int any_absent = 0;
      for (int64_t r = 0; r < b->n; r++) {
        idx[r] = -1;
        for (int c = 0; c < b->items[r].nvalues; c++) if (!strcmp(b->items[r].names->names[c], un[k])) idx[r] = c;  /* (a later duplicate wins, like AsMap) */
        if (idx[r] < 0) any_absent = 1;
      }
      fill_column(&cb->cols[k], b, f, un[k], k, 0, idx);
      if (any_absent) {
        cb->cols[k].absent = (uint8_t *)calloc((size_t)((b->n + 7) / 8), 1);
        for (int64_t r = 0; r < b->n; r++) if (idx[r] < 0) cb->cols[k].absent[r >> 3] |= (uint8_t)(1u << (r & 7));
      }
    }
    free(idx); free(un);
  } else
  for (int c = 0; c < cb->ncols; c++) fill_column(&cb->cols[c], b, f, f->names && c < f->names->n ? f->names->names[c] : "", c, 0, NULL);
  /* OldKeys: columns by the KeyNames of the first row that has them */
  const ora_item *fo = NULL;
  for (int64_t r = 0; r < b->n && !fo; r++) if (b->items[r].n_old > 0) fo = &b->items[r];
  if (fo) {
    cb->n_old_keys = fo->n_old;
    cb->old_keys = (tfgpu_column *)calloc((size_t)fo->n_old, sizeof(tfgpu_column));
    cb->old_keys_present = (uint8_t *)calloc((size_t)((b->n + 7) / 8), 1);
    for (int64_t r = 0; r < b->n; r++) if (b->items[r].n_old > 0) cb->old_keys_present[r >> 3] |= (uint8_t)(1u << (r & 7));
    for (int c = 0; c < fo->n_old; c++) fill_column(&cb->old_keys[c], b, f, fo->old_names->names[c], c, 1, NULL);
  }
  return cb;
}

void ora_columns_free(tfgpu_batch *cb) {
  if (!cb) return;
  for (int c = 0; c < cb->ncols; c++) {
    tfgpu_column *col = &cb->cols[c];
    free((void *)col->name); free(col->values); free(col->offsets); free(col->data); free(col->nanos); free(col->validity); free(col->absent);
  }
  for (int c = 0; c < cb->n_old_keys; c++) {
    tfgpu_column *col = &cb->old_keys[c];
    free((void *)col->name); free(col->values); free(col->offsets); free(col->data); free(col->nanos); free(col->validity);
  }
  free(cb->old_keys); free(cb->old_keys_present);
  free(cb->cols); free(cb->kind); free(cb->src_row); free(cb->part_id);
  free((void *)cb->table_ns); free((void *)cb->table_name);
  free(cb);
}

tfgpu_schema *ora_batch_schema(const ora_batch *b) { return ora_batch_item_schema(b, 0); }
const char *ora_batch_item_table(const ora_batch *b, int64_t row, int ns) { return row < b->n ? (ns ? b->items[row].ns : b->items[row].table) : ""; }
tfgpu_schema *ora_batch_item_schema(const ora_batch *b, int64_t row) {
  tfgpu_schema *s = (tfgpu_schema *)calloc(1, sizeof *s);
  if (row >= b->n || !b->items[row].schema) return s;
  const ora_schema *o = b->items[row].schema;
  s->ncols = o->ncols;
  s->cols = (tfgpu_colschema *)calloc((size_t)(o->ncols ? o->ncols : 1), sizeof(tfgpu_colschema));
  for (int i = 0; i < o->ncols; i++) {
    s->cols[i].name = dups(o->cols[i].name);
    s->cols[i].dtype = o->cols[i].dtype;
    s->cols[i].flags = (o->cols[i].key ? TFGPU_COL_KEY : 0) | (o->cols[i].required ? TFGPU_COL_REQUIRED : 0) | (o->cols[i].fake_key ? TFGPU_COL_FAKE_KEY : 0);
    s->cols[i].path = dups(o->cols[i].path);
    s->cols[i].original_type = dups(o->cols[i].original_type);
    s->cols[i].table_schema = o->cols[i].table_schema; s->cols[i].table_name = o->cols[i].table_name;  /* interned: not freed */
    s->cols[i].expression = o->cols[i].expression; s->cols[i].properties_json = o->cols[i].properties_json;
  }
  return s;
}
void ora_tschema_free(tfgpu_schema *s) {
  if (!s) return;
  for (int i = 0; i < s->ncols; i++) { free((void *)s->cols[i].name); free((void *)s->cols[i].path); free((void *)s->cols[i].original_type); }
  free(s->cols); free(s);
}
