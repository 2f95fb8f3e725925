/*
 * oracle/ora_srjson.c — CPU restatement of the Confluent Schema Registry parser's JSON-schema path.
 * TEST INFRASTRUCTURE ONLY (see ora.h).
 *
 *   ConfluentSrImpl.DoBatch / Do / DoBuf / DoOne   pkg/parsers/registry/confluentschemaregistry/engine/parser.go:108-152
 *   makeChangeItemsFromMessageWithJSON             engine/format_json.go:15-67
 *   processPayload / convertTypes                  engine/utils_json.go:27-128, types_json.go:25-32
 *   jsonx.NewDefaultDecoder(...).Decode(&map)      pkg/util/jsonx/json_decoder.go:9-14 = encoding/json Decoder + UseNumber
 *
 * encoding/json is the Go standard library (go1.22 semantics): the scanner grammar (scanner.go), Decoder.readValue's
 * end-of-value rules (stream.go), unquote (decode.go: escapes, surrogate pairs, invalid UTF-8 → U+FFFD), duplicate keys
 * (last one wins), json.Number, and json.Marshal of the decoded value for `any` columns (sorted keys, escapeHTML).
 * Pinned to the reference's canon: engine/gotest/canondata/result.json (TestClient) over testdata/test_raw_json_messages.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ora.h"

#include "ora_jv.h"
/* ---- frames: DoBuf / DoOne (parser.go:108-132) + the payload end of format_json.go:34-38 ---- */
typedef struct { tfgpu_sr_frame *f; int64_t n, cap; } framev;
static void frame_push(framev *v, tfgpu_sr_frame f) {
  if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 64; v->f = (tfgpu_sr_frame *)realloc(v->f, sizeof(tfgpu_sr_frame) * (size_t)v->cap); }
  v->f[v->n++] = f;
}
/* All frames a message would yield if no payload failed (a payload error ends its message: applied by the callers). */
static void message_frames(framev *out, const unsigned char *base, uint64_t a, uint64_t z, int64_t msg) {
  int32_t idx = 0;
  while (a < z) {
    tfgpu_sr_frame f; memset(&f, 0, sizeof f);
    f.msg = msg; f.index = idx++;
    if (z - a < 5) { f.start = a; f.len = (uint32_t)(z - a); f.code = TFGPU_ROW_SR_SHORT; frame_push(out, f); return; }
    if (base[a] != 0) { f.start = a; f.len = (uint32_t)(z - a); f.code = TFGPU_ROW_SR_MAGIC; frame_push(out, f); return; }
    f.schema_id = ((uint32_t)base[a + 1] << 24) | ((uint32_t)base[a + 2] << 16) | ((uint32_t)base[a + 3] << 8) | base[a + 4];
    a += 5;
    const unsigned char *zero = (const unsigned char *)memchr(base + a, 0, (size_t)(z - a));
    const uint64_t end = zero ? (uint64_t)(zero - base) : z;
    f.start = a; f.len = (uint32_t)(end - a); f.code = TFGPU_ROW_OK;
    frame_push(out, f);
    a = end;
  }
}
tfgpu_sr_frame *ora_sr_frames(const void *bytes, uint64_t len, const tfgpu_messages *msgs, int64_t *nframes) {
  framev v = {0};
  const int64_t nmsg = msgs ? msgs->nmsg : 1;
  for (int64_t m = 0; m < nmsg; m++) message_frames(&v, (const unsigned char *)bytes, msgs ? msgs->start[m] : 0, msgs ? msgs->start[m + 1] : len, m);
  *nframes = v.n;
  if (!v.f) v.f = (tfgpu_sr_frame *)malloc(sizeof(tfgpu_sr_frame));
  return v.f;
}

/* ---- one frame: processPayload + convertTypes ---- */
static int frame_item(ora_batch *out, const tfgpu_sr_json_options *o, ora_names *names, ora_schema *sch, const unsigned char *p, size_t n, int64_t ordinal) {
  int is_null = 0;
  jv *m = decode_map(p, n, &is_null);
  if (!m && !is_null) return TFGPU_ROW_JSON_SYNTAX;  /* "Can't unmarshal data changes from message" */
  ora_value *vals = (ora_value *)calloc((size_t)(o->nprops ? o->nprops : 1), sizeof(ora_value));
  char *listed = (char *)malloc((size_t)(o->nprops ? o->nprops : 1));
  memset(listed, 1, (size_t)(o->nprops ? o->nprops : 1));
  int code = 0;
  for (int j = 0; j < o->nprops && !code; j++) {
    const tfgpu_sr_property *pr = &o->props[j];
    const jv *v = NULL;
    if (m) { size_t kl = strlen(pr->name); for (int i = 0; i < m->nk; i++) if (m->klen[i] == kl && !memcmp(m->keys[i], pr->name, kl)) v = m->kids[i]; }
    ora_value *dst = &vals[j];
    if (!v) { if (pr->required) code = TFGPU_ROW_SR_REQUIRED; else if (o->is_generate_updates) listed[j] = 0; continue; }  /* absent optional field: nil (inserts carry every column); isGenerateUpdates: not listed (utils_json.go:57-63) */
    if (v->t == JV_NULL && !pr->required) continue;                         /* in == nil && nullable */
    switch (pr->json_type) {
      case TFGPU_SRT_BOOLEAN: if (v->t == JV_TRUE || v->t == JV_FALSE) { dst->kind = OV_BOOL; dst->v.b = v->t == JV_TRUE; } else code = TFGPU_ROW_SR_TYPE; break;
      case TFGPU_SRT_INTEGER: {
        int64_t x;
        if (v->t == JV_NUM && ora_parse_int(v->s, v->n, 10, 64, &x) == 0) { dst->kind = OV_I64; dst->v.i = x; } else code = TFGPU_ROW_SR_TYPE;  /* json.Number.Int64() */
        break;
      }
      case TFGPU_SRT_NUMBER: if (v->t == JV_NUM) { dst->kind = OV_JSONNUM; dst->slen = v->n; dst->s = (char *)malloc(v->n + 1); memcpy(dst->s, v->s, v->n + 1); } else code = TFGPU_ROW_SR_TYPE; break;
      case TFGPU_SRT_STRING: if (v->t == JV_STR) { dst->kind = OV_STRING; dst->slen = v->n; dst->s = (char *)malloc(v->n + 1); memcpy(dst->s, v->s, v->n + 1); } else code = TFGPU_ROW_SR_TYPE; break;
      default:  /* convertTypes default: the decoded value as it is */
        if (v->t != JV_NULL) { sbuf b = {0}; go_marshal(&b, v); dst->kind = OV_JSON; dst->s = b.p; dst->slen = b.n; }
    }
  }
  jv_free(m);
  if (code) { for (int j = 0; j < o->nprops; j++) ora_value_free(&vals[j]); free(vals); free(listed); return code; }
  ora_item *it = ora_batch_push(out);
  it->kind = o->is_generate_updates ? TFGPU_K_UPDATE : TFGPU_K_INSERT;  /* format_json.go:44-47 */
  it->ns = strdup(o->table_ns ? o->table_ns : ""); it->table = strdup(o->table_name ? o->table_name : ""); it->part_id = strdup("");
  it->schema = sch; sch->refs++;
  int nl = 0;
  for (int j = 0; j < o->nprops; j++) nl += listed[j];
  if (nl == o->nprops) { it->names = names; names->refs++; it->nvalues = o->nprops; it->values = vals; }
  else {  /* the item lists the fields its payload holds */
    ora_names *own = (ora_names *)calloc(1, sizeof *own);
    own->refs = 1; own->n = nl; own->names = (char **)calloc((size_t)(nl ? nl : 1), sizeof(char *));
    ora_value *v2 = (ora_value *)calloc((size_t)(nl ? nl : 1), sizeof(ora_value));
    int k = 0;
    for (int j = 0; j < o->nprops; j++) if (listed[j]) { own->names[k] = strdup(o->props[j].name); v2[k++] = vals[j]; }
    free(vals);
    it->names = own; it->nvalues = nl; it->values = v2;
  }
  free(listed);
  it->src_row = ordinal;
  return 0;
}

static int srt_dtype(int t) {
  switch (t) { case TFGPU_SRT_BOOLEAN: return TFGPU_T_BOOLEAN; case TFGPU_SRT_INTEGER: return TFGPU_T_INT64; case TFGPU_SRT_NUMBER: return TFGPU_T_FLOAT64;
               case TFGPU_SRT_STRING: return TFGPU_T_UTF8; default: return TFGPU_T_ANY; }
}

/* Rows for the frames of o->schema_id; errors (row = frame ordinal, code) for the frames the reference turns into
 * `_unparsed`; msg_of[r] (malloc'd, one per output row) = index of the row's message.                             */
ora_batch *ora_sr_json_parse(const tfgpu_sr_json_options *o, const void *bytes, uint64_t len, const tfgpu_messages *msgs, int64_t **msg_of) {
  ora_batch *out = ora_batch_new();
  ora_names *names = (ora_names *)calloc(1, sizeof *names);
  names->refs = 1; names->n = o->nprops; names->names = (char **)calloc((size_t)(o->nprops ? o->nprops : 1), sizeof(char *));
  tfgpu_colschema *cs = (tfgpu_colschema *)calloc((size_t)(o->nprops ? o->nprops : 1), sizeof *cs);
  for (int j = 0; j < o->nprops; j++) {
    names->names[j] = strdup(o->props[j].name);
    cs[j].name = o->props[j].name; cs[j].dtype = srt_dtype(o->props[j].json_type); cs[j].flags = o->props[j].required ? TFGPU_COL_REQUIRED : 0;
  }
  tfgpu_schema ts = {o->nprops, cs};
  ora_schema *sch = ora_schema_from(&ts);
  free(cs);
  const int64_t nmsg = msgs ? msgs->nmsg : 1;
  int64_t ordinal = 0, cap = 16, nrow = 0;
  int64_t *mo = (int64_t *)malloc(sizeof(int64_t) * (size_t)cap);
  for (int64_t m = 0; m < nmsg; m++) {
    framev fv = {0};
    message_frames(&fv, (const unsigned char *)bytes, msgs ? msgs->start[m] : 0, msgs ? msgs->start[m + 1] : len, m);
    int dead = 0;  /* an `_unparsed` item ends its message (doWithSchema returns nil) */
    for (int64_t k = 0; k < fv.n; k++, ordinal++) {
      const tfgpu_sr_frame *f = &fv.f[k];
      if (dead) continue;
      if (f->code) { dead = 1; if (o->report_frame_errors) ora_batch_add_error(out, ordinal, f->code, ""); continue; }
      if (f->schema_id != o->schema_id) continue;
      int code = frame_item(out, o, names, sch, (const unsigned char *)bytes + f->start, f->len, ordinal);
      if (code) { dead = 1; ora_batch_add_error(out, ordinal, code, ""); continue; }
      if (nrow == cap) { cap *= 2; mo = (int64_t *)realloc(mo, sizeof(int64_t) * (size_t)cap); }
      mo[nrow++] = m;
    }
    free(fv.f);
  }
  if (--names->refs == 0) { for (int j = 0; j < names->n; j++) free(names->names[j]); free(names->names); free(names); }
  ora_schema_unref(sch);
  if (msg_of) *msg_of = mo; else free(mo);
  return out;
}
