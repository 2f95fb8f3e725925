/*
 * oracle/ora.h — CPU restatement of the reference's transform/serialize hot
 * path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (transferia_amd/, libtfgpu.so) never does.
 *
 * The reference is Go and cannot be built here (no Go toolchain, ~330
 * un-vendored modules), so this is a restatement in plain C11 that keeps the
 * reference's data model — a batch is an array of row structs, each holding
 * an array of boxed, individually heap-allocated values (ChangeItem /
 * ColumnValues []interface{}, pkg/abstract/changeitem/change_item.go:27-80)
 * — and its scalar, per-row algorithms.  Every function cites the reference
 * file:line it follows.  Parity is pinned by the reference's own golden
 * vectors (tests/golden/, extracted by tools/extract_golden.py).
 */
#ifndef ORA_H
#define ORA_H
#include <stddef.h>
#include <stdint.h>
#include "../include/tfgpu.h" /* shared enums + the columnar struct used for test I/O */

#ifdef __cplusplus
extern "C" {
#endif

/* ---- boxed values (Go interface{} payloads) ---- */
typedef enum {
  OV_NIL = 0,
  OV_I8, OV_I16, OV_I32, OV_I64, OV_U8, OV_U16, OV_U32, OV_U64,
  OV_F32, OV_F64, OV_BOOL,
  OV_STRING, OV_BYTES, OV_JSONNUM, OV_JSON,
  OV_TIME, OV_DURATION
} ora_vkind;

typedef struct ora_value {
  int kind;
  union {
    int64_t i;
    uint64_t u;
    double f64;
    float f32;
    int b;
    struct { int64_t sec; int32_t nsec; } t;
  } v;
  char *s;      /* heap payload for STRING/BYTES/JSONNUM/JSON */
  size_t slen;
} ora_value;

typedef struct ora_colschema {
  char *name;
  int dtype;          /* tfgpu_dtype */
  int key;
  char *path;
  char *original_type;
  /* the ColSchema fields only the wire form reads (col_schema.go:14-29): interned strings (ora_intern), never freed,
   * so that the struct copies all over the oracle may share them */
  const char *table_schema, *table_name, *expression, *properties_json;
  int fake_key, required;
} ora_colschema;
const char *ora_intern(const char *s); /* NULL → "" */

typedef struct ora_schema {
  int ncols;
  ora_colschema *cols;
  int refs;
} ora_schema;

typedef struct ora_names { int n; char **names; int refs; } ora_names;

/* ChangeItem (row-level subset that the path reads or writes) */
typedef struct ora_item {
  int kind;            /* tfgpu_kind */
  char *ns;            /* Schema */
  char *table;         /* Table  */
  char *part_id;       /* PartID */
  ora_names *names;    /* ColumnNames (shared, write-once) */
  int nvalues;
  ora_value *values;   /* ColumnValues */
  ora_schema *schema;  /* TableSchema (shared pointer, replaced not mutated) */
  int64_t src_row;     /* index in the original input batch */
  /* OldKeys (old_keys.go:3-7): KeyNames (shared) / KeyValues; n_old = len(KeyValues), 0 = none */
  ora_names *old_names;
  int n_old;
  ora_value *old_values;
} ora_item;

typedef struct ora_error { int64_t row; int code; char *msg; } ora_error;

typedef struct ora_batch {
  int64_t n, cap;
  ora_item *items;
  int64_t nerr, errcap;
  ora_error *errs;
} ora_batch;

/* ---- memory / construction ---- */
ora_batch *ora_batch_new(void);
void ora_batch_free(ora_batch *b);
void ora_value_free(ora_value *v);
ora_value ora_value_clone(const ora_value *v);
ora_schema *ora_schema_from(const tfgpu_schema *s);
void ora_schema_unref(ora_schema *s);
ora_item *ora_batch_push(ora_batch *b); /* zeroed item */
void ora_batch_add_error(ora_batch *b, int64_t row, int code, const char *msg);
void ora_item_clear(ora_item *it);

/* columnar <-> row conversion (test I/O only; not timed) */
ora_batch *ora_from_columns(const tfgpu_batch *cb, const tfgpu_schema *schema);
/* Allocates a host tfgpu_batch with malloc; all rows must share column layout. */
tfgpu_batch *ora_to_columns(const ora_batch *b);
void ora_columns_free(tfgpu_batch *cb);
tfgpu_schema *ora_batch_schema(const ora_batch *b); /* schema of first row, malloc'd */
tfgpu_schema *ora_batch_item_schema(const ora_batch *b, int64_t row); /* TableSchema of one item, malloc'd */
const char *ora_batch_item_table(const ora_batch *b, int64_t row, int ns); /* Table, or Schema with ns != 0 */
void ora_tschema_free(tfgpu_schema *s);

/* ---- Go stdlib behaviours restated (strconv/fmt/time) ---- */
size_t ora_fmt_int(char *dst, int64_t v);
size_t ora_fmt_uint(char *dst, uint64_t v);
/* strconv.FormatFloat(f, fmt, -1, bits), fmt in {'g'(=fmt %v), 'f'} */
size_t ora_fmt_float(char *dst, double f, char fmt, int bits);
size_t ora_json_float(char *dst, double f, int bits);
size_t ora_fmt_date(char *dst, int64_t sec);               /* time.DateOnly, UTC */
size_t ora_fmt_rfc3339nano(char *dst, int64_t sec, int32_t nsec); /* UTC */
size_t ora_fmt_time_string(char *dst, int64_t sec, int32_t nsec); /* Time.String(), UTC */
size_t ora_fmt_duration(char *dst, int64_t ns);            /* Duration.String() */
/* strconv.ParseInt(s, base, bits): 0 ok, 1 syntax error, 2 range error */
int ora_parse_int(const char *s, size_t n, int base, int bits, int64_t *out);
int ora_parse_uint(const char *s, size_t n, int base, int bits, uint64_t *out);
int ora_parse_float(const char *s, size_t n, int bits, double *out); /* strconv.ParseFloat */
int ora_parse_duration(const char *s, size_t n, int64_t *out);              /* time.ParseDuration: 0 ok, 1 error */
int ora_cast_string_to_duration(const char *s, size_t n, int64_t *out);     /* cast.ToDurationE(string) */
int ora_parse_bool(const char *s, size_t n, int *out);     /* strconv.ParseBool */
/* time.Parse(layout, s) for the layouts used on the path; returns 0 on success */
int ora_time_parse(const char *layout, const char *s, size_t n, int64_t *sec, int32_t *nsec);
void ora_civil_from_days(int64_t z, int64_t *y, int *m, int *d);
int64_t ora_days_from_civil(int64_t y, int m, int d);

/* ---- hashes ---- */
void ora_sha256(const void *data, size_t n, uint8_t out[32]);
void ora_hmac_sha256(const void *key, size_t klen, const void *msg, size_t mlen, uint8_t out[32]);
uint32_t ora_crc32_ieee(const void *data, size_t n);
uint32_t ora_fnv1a32(const void *data, size_t n);

/* ---- a5: SerializeToString (to_string.go:145-178) ---- */
/* returns malloc'd string, *len set */
char *ora_serialize_to_string(const ora_value *v, int dtype, size_t *len);

/* ---- transformers: Apply(type, config_json) ---- */
typedef struct ora_transformer ora_transformer;
ora_transformer *ora_transformer_new(const char *type_name, const char *config_json, char *err, size_t errcap);
void ora_transformer_free(ora_transformer *t);
int ora_transformer_suitable(const ora_transformer *t, const char *ns, const char *table, const tfgpu_schema *s);
tfgpu_schema *ora_transformer_result_schema(const ora_transformer *t, const tfgpu_schema *s);
/* Consumes `in` rows (ownership moves like AsyncPush), returns new batch with
 * Transformed rows and Errors. */
ora_batch *ora_transformer_apply(const ora_transformer *t, ora_batch *in);

/* filter grammar (library/go/yandex/cloud/filter) exposed for tests:
 * returns number of terms or -1 and fills err */
int ora_filter_parse_check(const char *expr, char *err, size_t errcap);

/* ---- CSV: pkg/csv.Reader + s3 CSVReader ---- */
/* csv.Reader.ReadAll over a buffer; returns fields as a flat malloc'd table.
 * out_rows[i] = number of fields of line i (or -1 for a nil line), fields in
 * order in out_fields (each malloc'd, NUL-terminated, with length array).     */
typedef struct ora_csv_table {
  int64_t nlines;
  int32_t *nfields;    /* -1: ReadLine returned (nil,nil) */
  int32_t *line_err;   /* tfgpu_rowerr for that line, 0 ok */
  int64_t ntotal;
  char **fields;
  size_t *lens;
  uint64_t consumed;
} ora_csv_table;
ora_csv_table *ora_csv_read_all(const tfgpu_csv_options *o, const void *bytes, uint64_t len);
void ora_csv_table_free(ora_csv_table *t);
/* parseCSVRows + doParse (constructCI + Strictify) → batch of typed rows */
ora_batch *ora_csv_parse(const tfgpu_csv_options *o, const tfgpu_schema *schema, const char *ns,
                         const char *table, const void *bytes, uint64_t len, uint64_t *consumed);

int64_t ora_csv_split_rows(const void *bytes, uint64_t len, uint64_t **ends); /* csv.Splitter (pkg/csv/splitter.go) */
/* getCorrespondingValue alone (reader_csv.go:345-452), for the reference's TestParse*Value tables */
ora_batch *ora_strictify(ora_batch *b); /* strictify.Strictify over every item, in place; failing items are reported and left as they were */
ora_batch *ora_csv_corresponding_value(const tfgpu_csv_options *o, const char *s, uint64_t n, int dtype);

/* ---- a17: generic JSON parser (pkg/parsers/generic/generic_parser.go) — ora_jsonparse.c ---- */
typedef struct ora_json_lines ora_json_lines; /* per non-empty line: status / code / column / msg / idx / row */
ora_batch *ora_json_parse(const tfgpu_json_options *o, const tfgpu_schema *fields, const void *bytes, uint64_t len,
                          const tfgpu_messages *msgs, ora_json_lines **lines_out);
void ora_json_lines_free(ora_json_lines *l);
tfgpu_schema *ora_json_result_schema(const tfgpu_json_options *o, const tfgpu_schema *fields);
int ora_batch_value(const ora_batch *b, int64_t row, int col, int *kind, int64_t *i64, double *f64, const char **s, size_t *slen, int32_t *nsec);
double ora_fastfloat_parse_best_effort(const char *s, size_t n);       /* fastjson/fastfloat v1.6.4 */
int64_t ora_fastfloat_parse_int64_best_effort(const char *s, size_t n);
uint64_t ora_fastfloat_parse_uint64_best_effort(const char *s, size_t n);

/* ---- serializers ---- */
/* format = TFGPU_FMT_*; returns malloc'd bytes */
char *ora_serialize(int format, const ora_batch *b, uint64_t *len);
/* ---- a24: abstract.Collapse (change_item_collapse.go:48-134); consumes nothing, returns a new batch ---- */
ora_batch *ora_collapse(const ora_batch *in);
int ora_item_keys_changed(const ora_item *c);               /* ChangeItem.KeysChanged, change_item.go:237-286 */
void ora_keys_changed(const ora_batch *b, uint8_t *out);    /* per item */
ora_batch *ora_split_updated_pkeys(const ora_batch *in, int64_t **lens_out, int64_t *nlists); /* utils.go:75-128 */
ora_batch *ora_batch_from_json(const char *text);  /* row-wise test input (items with differing ColumnNames) */
int ora_batch_item_info(const ora_batch *b, int64_t row, int *kind, int *nvalues, int *n_old, int64_t *src_row);
const char *ora_batch_item_name(const ora_batch *b, int64_t row, int col, int old);
int ora_batch_old_value(const ora_batch *b, int64_t row, int col, int *kind, int64_t *i64, const char **s, size_t *slen);
int64_t ora_batch_len(const ora_batch *b);
char *ora_serialize_ex(int format, const ora_batch *b, const tfgpu_serialize_options *opts, uint64_t *len);
/* §8f.4 queue serializers (pkg/serializer/queue): message values back to back, NULL = the reference's Serialize
 * fails (or a value form is outside the restatement).  msg_start / msg_row are malloc'd with *nmsg + 1 entries. */
char *ora_queue_serialize(const tfgpu_queue_options *o, const ora_batch *b, const tfgpu_row_meta *m, uint64_t *len,
                          uint64_t **msg_start, int64_t **msg_row, int64_t *nmsg);

/* §8 f1: DebeziumImpl.DoBatch without a schema registry (ora_debezium.c) */
ora_batch *ora_debezium_parse(const void *bytes, uint64_t len, const tfgpu_messages *msgs, int32_t **code, int64_t **row_of, uint32_t **id, uint64_t **lsn,
                              uint64_t **commit_time, uint8_t **names_form);

/* §8 f2: marshalChangeItemInto + the driver's Native block layout (ora_chnative.c); NULL = refused */
char *ora_ch_native_block(const ora_batch *rows, const char *const *names, const char *const *types, int ncols, uint64_t *len);

/* a23: util.DeepSizeof(ColumnValues) per item (ora_sizeof.c); flags & 1: JSON numbers decoded as float64 */
uint64_t ora_deepsizeof(const ora_batch *b, int flags, uint64_t *per_row);
uint64_t ora_deepsizeof_value(const ora_value *v, int flags);

/* §8f.1 Confluent Schema Registry parser, JSON schemas (ora_srjson.c) */
tfgpu_sr_frame *ora_sr_frames(const void *bytes, uint64_t len, const tfgpu_messages *msgs, int64_t *nframes); /* malloc'd */
ora_batch *ora_sr_json_parse(const tfgpu_sr_json_options *o, const void *bytes, uint64_t len, const tfgpu_messages *msgs, int64_t **msg_of);

#ifdef __cplusplus
}
#endif
/* lookupComplex over a top-level string value (ora_lookup.c; parsers/generic/lookup.go:10-59).  segs[0] is the top-level key
 * (already resolved to `top`), segs[1..] the nested field names. */
enum { ORA_LOOKUP_STRING = 0, ORA_LOOKUP_ERROR = 1, ORA_LOOKUP_NIL = 2, ORA_LOOKUP_OTHER = 3 };
int ora_lookup_complex(const char *top, size_t topn, const char *const *segs, int nsegs, char **out, size_t *outn);

#endif
