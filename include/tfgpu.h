/*
 * tfgpu.h — C ABI of the MI355X-native columnar transform stage for Transferia.
 *
 * This is the drop-in boundary.  The reference has no FFI of
 * its own (it is 100 % Go); the entry points below are what a cgo shim that
 * implements the reference's plugin interfaces would bind:
 *
 *   abstract.Transformer{Type,Description,Suitable,ResultSchema,Apply}
 *       /root/reference/pkg/abstract/transformer.go:32-38
 *   transformer.Register / transformer.New (config arrives as JSON-remapped map)
 *       /root/reference/pkg/transformer/registry.go:34-47
 *   parsers.Parser{Do,DoBatch}            pkg/parsers/abstract.go:35-38
 *   serializer.BatchSerializer            pkg/serializer/interface.go:11-26
 *   s3 CSVReader.Read / parseCSVRows      pkg/providers/s3/reader/registry/csv/reader_csv.go:85-247
 *   httpuploader.MarshalCItoJSON          pkg/providers/clickhouse/httpuploader/marshal.go:82-125
 *
 * Conventions
 *   - every function returns a tfgpu_status (0 = ok) unless stated otherwise;
 *     tfgpu_last_error() returns a thread-local human-readable message.
 *   - plain pointers and sizes only; no C++/torch types cross this boundary.
 *   - plans are immutable after creation and safe for concurrent use
 *     (transformation.Push calls Apply from one goroutine per table,
 *     pkg/transformer/transformation.go:131-135).
 *   - the library never retains caller pointers after a call returns.
 *   - there is NO CPU fallback: without a gfx950 device every compute entry
 *     point fails with TFGPU_ERR_DEVICE.
 */
#ifndef TFGPU_H
#define TFGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: tfgpu_column gained `absent` (its size and array stride grew by one pointer) and tfgpu_batch gained `col_order` at its tail.
 * A binding compiled against version 1's structs MUST NOT call this library: tfgpu_abi_version() is the first call a binding makes,
 * and it refuses to go on when the number differs from the TFGPU_ABI_VERSION it was compiled with. */
#define TFGPU_ABI_VERSION 2

typedef enum tfgpu_status {
  TFGPU_OK = 0,
  TFGPU_ERR_INVALID = 1,     /* bad argument / malformed batch                 */
  TFGPU_ERR_CONFIG = 2,      /* transformer config rejected (factory error)    */
  TFGPU_ERR_UNSUPPORTED = 3, /* valid in the reference, not device-resident    */
  TFGPU_ERR_DEVICE = 4,      /* HIP error / no gfx950 device                   */
  TFGPU_ERR_NOMEM = 5,
  TFGPU_ERR_UNKNOWN_TYPE = 6 /* transformer type not registered                */
} tfgpu_status;

/* ColSchema.DataType — the YT type names the reference uses as strings
 * (pkg/abstract/typesystem/schema.go:49-69).
 * NB: "string" is arbitrary bytes (ytschema.TypeBytes), "utf8" is text. */
typedef enum tfgpu_dtype {
  TFGPU_T_INVALID = 0,
  TFGPU_T_INT8 = 1, TFGPU_T_INT16, TFGPU_T_INT32, TFGPU_T_INT64,
  TFGPU_T_UINT8, TFGPU_T_UINT16, TFGPU_T_UINT32, TFGPU_T_UINT64,
  TFGPU_T_FLOAT32,   /* "float"     */
  TFGPU_T_FLOAT64,   /* "double"    */
  TFGPU_T_BOOLEAN,   /* "boolean"   */
  TFGPU_T_BYTES,     /* "string"    */
  TFGPU_T_UTF8,      /* "utf8"      */
  TFGPU_T_DATE,      /* "date"      */
  TFGPU_T_DATETIME,  /* "datetime"  */
  TFGPU_T_TIMESTAMP, /* "timestamp" */
  TFGPU_T_INTERVAL,  /* "interval"  */
  TFGPU_T_ANY,       /* "any"       */
  TFGPU_T__COUNT
} tfgpu_dtype;

/* Physical representation of a column = the Go dynamic type of the boxed
 * ColumnValues it was fanned out from.  The reference dispatches on the
 * dynamic type in many places (filter_rows.go:180-365, to_string.go:145-171),
 * so it is part of the data, separate from the schema DataType. */
typedef enum tfgpu_repr {
  TFGPU_R_INVALID = 0,
  TFGPU_R_INT8 = 1, TFGPU_R_INT16, TFGPU_R_INT32, TFGPU_R_INT64,
  TFGPU_R_UINT8, TFGPU_R_UINT16, TFGPU_R_UINT32, TFGPU_R_UINT64,
  TFGPU_R_FLOAT32, TFGPU_R_FLOAT64,
  TFGPU_R_BOOL,     /* 1 byte per value, 0/1                                  */
  TFGPU_R_STRING,   /* Go string      : offsets[nrows+1] (u32) + data         */
  TFGPU_R_BYTES,    /* Go []byte      : offsets + data                        */
  TFGPU_R_JSONNUM,  /* json.Number    : offsets + data (decimal text)         */
  TFGPU_R_JSON,     /* any            : offsets + data (json.Marshal text)    */
  TFGPU_R_TIME,     /* time.Time (UTC): values = i64 unix seconds,
                                        nanos  = i32 nanoseconds or NULL (=0) */
  TFGPU_R_DURATION, /* time.Duration  : values = i64 nanoseconds              */
  TFGPU_R__COUNT
} tfgpu_repr;

/* ChangeItem.Kind (pkg/abstract/changeitem/kind.go:5-44), row-level subset.
 * filter_rows treats "Insert"/"insert" etc. as equal (filter_rows.go:92-96). */
typedef enum tfgpu_kind {
  TFGPU_K_INSERT = 0, TFGPU_K_UPDATE = 1, TFGPU_K_DELETE = 2,
  TFGPU_K_OTHER = 3, /* any non-row kind: passes through every row kernel */
  TFGPU_K_SYNCHRONIZE = 4 /* SynchronizeKind: a non-row kind too, but InsertsOnly counts it as an insert (change_item_collapse.go:37-44) */
} tfgpu_kind;

enum { TFGPU_MEM_HOST = 0, TFGPU_MEM_DEVICE = 1 };

enum { TFGPU_COL_KEY = 1u /* ColSchema.PrimaryKey */, TFGPU_COL_REQUIRED = 2u, TFGPU_COL_FAKE_KEY = 4u /* ColSchema.FakeKey */ };

/* One ColSchema (pkg/abstract/changeitem/col_schema.go:14-29). */
typedef struct tfgpu_colschema {
  const char *name;          /* ColumnName                                    */
  int32_t dtype;             /* tfgpu_dtype                                   */
  uint32_t flags;            /* TFGPU_COL_*                                   */
  const char *path;          /* ColSchema.Path (CSV: decimal column index)    */
  const char *original_type; /* ColSchema.OriginalType or NULL                */
  /* the fields only the wire form reads (ChangeItem.MarshalJSON → NativeSerializer); NULL = ""    */
  const char *table_schema;  /* ColSchema.TableSchema                         */
  const char *table_name;    /* ColSchema.TableName                           */
  const char *expression;    /* ColSchema.Expression                          */
  const char *properties_json; /* json.Marshal(ColSchema.Properties) when the map is non-empty (the key is omitempty), else NULL */
} tfgpu_colschema;

typedef struct tfgpu_schema {
  int32_t ncols;
  tfgpu_colschema *cols;
} tfgpu_schema;

/* One column of a batch: ColumnNames[i] + all rows' ColumnValues[i]. */
typedef struct tfgpu_column {
  const char *name;
  int32_t dtype;      /* schema DataType looked up BY NAME       */
  int32_t repr;       /* tfgpu_repr                                           */
  void *values;       /* fixed-width reprs: nrows elements; else NULL         */
  uint32_t *offsets;  /* var-width reprs: nrows+1 offsets into data           */
  uint8_t *data;      /* var-width payload                                    */
  uint64_t data_len;  /* bytes in data                                        */
  int32_t *nanos;     /* TFGPU_R_TIME only; NULL = all zero                   */
  uint8_t *validity;  /* bitmap, bit i (LSB first) = 1 if value i != nil;
                         NULL = no nil values                                 */
  uint8_t *absent;    /* bitmap, bit i = 1: row i's ColumnNames do NOT list this
                         column at all (an Update that leaves a TOASTed column
                         out, pkg/abstract/changeitem/change_item_collapse.go:7-35;
                         the Debezium receiver's `__debezium_unavailable_value`,
                         pkg/debezium/receiver_engine.go:143-148) — a third cell
                         state beside a value and nil.  An ABSENT cell also reads
                         nil (its validity bit is 0).  NULL = every row lists the
                         column (the uniform case).  A row's ColumnNames are the
                         batch's columns it is not absent from, in batch order.
                         Written by tfgpu_batch_upload and the Debezium
                         receiver (`__debezium_unavailable_value`); read by
                         tfgpu_collapse, tfgpu_keys_changed, sharder_transformer,
                         tfgpu_partition, tfgpu_exchange, view / download, the
                         native queue format and the Debezium emitter; every
                         other entry computes on VALUES and refuses such a batch
                         by name (TFGPU_ERR_UNSUPPORTED).                       */
} tfgpu_column;

/* A batch = one contiguous same-table, same-schema run of ChangeItems, which
 * is exactly what transformation.do hands to Apply (transformation.go:252-257). */
typedef struct tfgpu_batch {
  int64_t nrows;
  int32_t ncols;
  tfgpu_column *cols;
  const char *table_ns;   /* ChangeItem.Schema */
  const char *table_name; /* ChangeItem.Table  */
  uint8_t *kind;          /* tfgpu_kind per row; NULL = all insert            */
  int32_t *src_row;       /* outputs: index of the input row each output row
                             came from (the fan-in key for row meta);
                             NULL on input = identity                         */
  uint32_t *part_id;      /* outputs of sharder_transformer: PartID = itoa()  */
  int32_t mem;            /* TFGPU_MEM_HOST / TFGPU_MEM_DEVICE                */
  /* ChangeItem.OldKeys (pkg/abstract/changeitem/old_keys.go:3-7) of Update / Delete rows: one column per
   * OldKeys.KeyNames entry (name = the key name, values = KeyValues, validity = nil), same row count as the batch.
   * old_keys_present: bitmap, bit r = len(items[r].OldKeys.KeyValues) > 0; NULL with n_old_keys > 0 = every row.
   * Only tfgpu_collapse reads them; every row-moving step (filter, partition, …) carries them along.          */
  int32_t n_old_keys;
  tfgpu_column *old_keys;
  uint8_t *old_keys_present;
  const tfgpu_schema *schema; /* ChangeItem.TableSchema of the run (columns in schema
                             order), or NULL = the batch columns in their order.
                             Transformers that walk the SCHEMA and look values up
                             by name (sharder.go:134-143) need it when ColumnNames
                             and TableSchema differ.               */
  uint16_t *col_order;     /* [nrows][ncols], row-major, or NULL = every row lists its columns in batch order.  Row r's
                              ColumnNames are cols[col_order[r*ncols + 0]], cols[col_order[r*ncols + 1]], … for the columns it
                              lists (tfgpu_column.absent), followed by the ones it does not (in no particular order): a full
                              permutation of 0 .. ncols-1.  An OUTPUT of tfgpu_collapse only: compareColumns appends the names a
                              later Update brings to the END of the merged row's list (change_item_collapse.go:24-33), which
                              batch order cannot say.  Read by view / download (the fan-in rebuilds ColumnNames from it), the
                              native queue format (names and values in that order) and the Debezium emitter (a map: order does
                              not matter); everything else refuses such a batch by name; tfgpu_batch_upload refuses it too.   */
} tfgpu_batch;

/* Per-row failure, the C image of abstract.TransformerError
 * (pkg/abstract/transformer.go:40-48): the row is omitted from the output. */
typedef enum tfgpu_rowerr {
  TFGPU_ROW_OK = 0,
  TFGPU_ROW_UNSUPPORTED_KIND = 1,  /* filter_rows.go:103-107 (fatal)          */
  TFGPU_ROW_COLUMN_NOT_FOUND = 2,  /* filter_rows.go:149-154 (fatal)          */
  TFGPU_ROW_INT_OVERFLOW = 3,      /* filter_rows/util.go:65-68               */
  TFGPU_ROW_TYPE_PAIR = 4,         /* "Unsupported type pair" filter_rows:364 */
  TFGPU_ROW_MISSING_CELL = 5,      /* reader_csv.go:303-316                   */
  TFGPU_ROW_CAST = 6,              /* strictify cast error                    */
  TFGPU_ROW_RANGE = 7,             /* StrictifyRangeError                     */
  TFGPU_ROW_QUOTE = 8,             /* csv: element is a single quote char     */
  TFGPU_ROW_DOUBLE_QUOTE = 9,      /* csv: errDoubleQuotesDisabled            */
  TFGPU_ROW_QUOTING_DISABLED = 10, /* csv: errQuotingDisabled                 */
  TFGPU_ROW_HOST_FALLBACK = 11,    /* value form not handled on device (free-
                                      form dates …): caller must run this row
                                      through the stock Go path               */
  TFGPU_ROW_JSON_SYNTAX = 12,      /* generic parser: Unmarshal(line) failed →
                                      NewUnparsed (generic_parser.go:545-550)  */
  TFGPU_ROW_PARSE_VAL = 13,        /* ParseVal error on a key/required column →
                                      newUnparsed (generic_parser.go:363-367)  */
  TFGPU_ROW_NIL_KEY = 14,          /* nil in a key/required column →
                                      newUnparsed (generic_parser.go:370-372)  */
  TFGPU_ROW_SR_SHORT = 15,         /* confluent SR: message shorter than the 5-byte wire prefix
                                      (confluentschemaregistry/engine/parser.go:109-112)          */
  TFGPU_ROW_SR_MAGIC = 16,         /* confluent SR: first byte is not 0 (parser.go:113-116)       */
  TFGPU_ROW_SR_TYPE = 17,          /* confluent SR json: convertTypes "wrong type" / Number.Int64
                                      error (utils_json.go:97-128)                                */
  TFGPU_ROW_SR_REQUIRED = 18,      /* confluent SR json: required field absent (utils_json.go:54-56) */
  /* debezium parser (pkg/parsers/registry/debezium/engine/parser.go:33-57): why a message became an `_unparsed` item   */
  TFGPU_ROW_DBZ_UNPACK = 19,       /* empty message, or IncludeSchema.Unpack's json.Unmarshal fails (include_schema.go:13-25) */
  TFGPU_ROW_DBZ_PAYLOAD = 20,      /* UnmarshalPayload fails: no payload, or a Payload / Source field of the wrong JSON type
                                      (receiver.go:151-154, debezium_schema.go:31-56)                                   */
  TFGPU_ROW_DBZ_OP = 21,           /* opToKind: unknown op (kind.go:34-46)                                              */
  TFGPU_ROW_DBZ_SCHEMA = 22,       /* receiveSchema fails: the schema does not unmarshal, or a field's Kafka type has no
                                      receiver (receiver.go:60-96, receiver_engine.go:108-146)                          */
  TFGPU_ROW_SR_PROTO = 25,         /* confluent SR protobuf: the schema text does not compile, or the message bytes do not unmarshal (format_protobuf.go:44-53) */
  TFGPU_ROW_DROPPED = 24,          /* registry-framed Debezium: an earlier event of the same Kafka message failed; DoBuf stops there */
  TFGPU_ROW_DBZ_FIELD = 23         /* a schema field is missing from before / after, or receiveField rejects its value
                                      (receiver.go:216-230, receiver_engine.go:148-287)                                 */
} tfgpu_rowerr;

typedef struct tfgpu_row_error {
  int64_t row;      /* index into the INPUT batch / line number for parsers   */
  int32_t code;     /* tfgpu_rowerr                                           */
  int32_t step;     /* index of the plan in the chain that raised it          */
  int32_t column;   /* column index the error refers to, or -1                */
} tfgpu_row_error;

typedef struct tfgpu_plan tfgpu_plan;     /* one transformer instance         */
typedef struct tfgpu_dbatch tfgpu_dbatch; /* a batch resident in HBM          */
typedef struct tfgpu_dbuf tfgpu_dbuf;     /* a byte buffer resident in HBM    */

/* ---- library ----------------------------------------------------------- */
int tfgpu_abi_version(void);
const char *tfgpu_last_error(void);
/* Bind the calling process to HIP device `device` (one process per GPU).    */
int tfgpu_init(int device);
/* Bind the process to SEVERAL devices: lane k (below) lives on devices[k mod ndevices], so one worker process — one Go
 * worker with a goroutine per lane — drives every GPU of the node with a row-range shard each (tfgpu_shard_rows), the
 * shape SURVEY §8(e) gives the path: transformation.go:131-135 is one goroutine per table over independent rows.
 * tfgpu_init(d) is the list {d}.  A second call with another list fails until tfgpu_shutdown.                       */
int tfgpu_init_devices(const int *devices, int ndevices);
int tfgpu_shutdown(void);
int tfgpu_device_count(int *out);
int tfgpu_synchronize(void);
/* The HIP stream every kernel of this library is launched on (hipStream_t). */
void *tfgpu_stream(void);

/* Lanes — the device half of the parsequeue (pkg/parsequeue/parsequeue.go:57-154: bounded-parallel
 * parse, in-order push).  A lane is an independent (HIP stream, HBM block cache, pinned ring) set of the
 * process's GPU; a host thread (goroutine locked to its OS thread) binds itself to lane k and every call it
 * makes is enqueued there, so work on different lanes overlaps on the device: H2D of chunk N+1 beside the
 * kernels of chunk N beside the D2H of chunk N-1.  Lane 0 exists after tfgpu_init and is every thread's
 * default.  Handles (dbatch, dbuf) must be used and freed on the lane that made them.  With a device
 * list (tfgpu_init_devices) lane k is a lane of device k mod ndevices.                                */
int tfgpu_lane_count(void);        /* how many lanes may be used (0 .. count-1)                          */
int tfgpu_lane_use(int lane);      /* bind the calling thread; creates the lane on first use             */
int tfgpu_lane_current(void);
int tfgpu_lane_device(int lane, int *device);  /* the HIP device lane `lane` lives (or will live) on            */

/* Row-range shards of a device batch and their ordered merge (tf_shard.hip) — how one process spreads a batch over the
 * lanes of several devices.  Nothing on the path looks across rows (transformation.go:131-135, filter / mask / cast are
 * per row), so the shards need no exchange; abstract.Collapse is the exception and has tfgpu_exchange.
 *   tfgpu_dbatch_slice:   rows [row0, row0 + nrows) as a batch of the calling lane; row0 must be a multiple of 8.
 *   tfgpu_dbatch_to_lane: `b` as a batch of lane `lane`: a deep copy (hipMemcpyAsync over xGMI / PCIe) when that lane is
 *                         on another device, shared immutable buffers otherwise.  `b` must be complete on its own lane.
 *   tfgpu_shard_rows:     nshards balanced row ranges cut at multiples of 64 rows; out[g] lives on lane lanes[g] (lanes
 *                         == NULL: lane g); row0[g] (optional) = first row of shard g in `b`.  Called on b's lane.
 *   tfgpu_dbatch_concat:  the parts' rows, parts in order, on the calling lane.  The parts may live on any lane / device
 *                         (their lanes synchronised by the caller); columns, representations, TableSchema and table id
 *                         must agree.  row_base[g] (optional) is added to part g's src_row (a part without src_row
 *                         counts 0, 1, …), which turns shard-local source rows back into rows of the sharded batch.    */
int tfgpu_dbatch_slice(const tfgpu_dbatch *b, int64_t row0, int64_t nrows, tfgpu_dbatch **out);
int tfgpu_dbatch_to_lane(const tfgpu_dbatch *b, int lane, tfgpu_dbatch **out);
int tfgpu_shard_rows(const tfgpu_dbatch *b, int nshards, const int *lanes, tfgpu_dbatch **out, int64_t *row0);
int tfgpu_dbatch_concat(const tfgpu_dbatch *const *parts, int nparts, const int64_t *row_base, tfgpu_dbatch **out);

/* ---- ParseQueue and Bufferer: the scheduling either side of the device path (tf_pipeline.cpp) ----------------------------
 * tfgpu_parsequeue = parsequeue.ParseQueue[TData] (pkg/parsequeue/parsequeue.go:16-217): bounded-parallel parse, in-order
 * push, in-order ack.  `msg` is the caller's handle of a message batch (TData).  Add() starts parse(user, msg, slot, &parsed) at
 * once on a thread of its own and blocks while `parallelism` parses are in flight; slot is in [0, parallelism) and unique among
 * the parses running at the same time — the shim binds it to a device lane (tfgpu_lane_use(1 + slot % lanes)), so parses
 * overlap on the GPU.  In Add order: push(user, parsed, &ticket) = sink.AsyncPush (must not block on the sink), then
 * wait(user, ticket, timeout_ms) = the read of its error channel (0 = pushed, TFGPU_PQ_PENDING = not yet, else the error), then
 * ack(user, msg, push_start_ns).  The first error of a parse, a push or an ack cancels the queue; Add then fails and
 * tfgpu_parsequeue_error returns that error's code and "parse queue: <stage> error: …".  parallelism 0 = 10, below 2 = 2, as
 * parsequeue.New.  Close cancels and joins; it does not wait for pending pushes (the reference's Close does not either).
 * A parse result the queue never hands to push() — the queue was cancelled by an error, or closed with parses in flight — goes
 * to release(user, parsed) during Close, when one is set (tfgpu_parsequeue_set_release); without it the shim owns those handles. */
#define TFGPU_PQ_PENDING (-1)
typedef struct tfgpu_parsequeue tfgpu_parsequeue;
typedef int (*tfgpu_pq_parse_fn)(void *user, uint64_t msg, int slot, void **parsed);
typedef int (*tfgpu_pq_push_fn)(void *user, void *parsed, uint64_t *ticket);
typedef int (*tfgpu_pq_wait_fn)(void *user, uint64_t ticket, int64_t timeout_ms);
typedef int (*tfgpu_pq_ack_fn)(void *user, uint64_t msg, int64_t push_start_ns);
typedef void (*tfgpu_pq_release_fn)(void *user, void *parsed);
int tfgpu_parsequeue_create(int parallelism, tfgpu_pq_parse_fn parse, tfgpu_pq_push_fn push, tfgpu_pq_wait_fn wait, tfgpu_pq_ack_fn ack, void *user, tfgpu_parsequeue **out);
int tfgpu_parsequeue_add(tfgpu_parsequeue *q, uint64_t msg);
int tfgpu_parsequeue_error(tfgpu_parsequeue *q, char *msg, size_t cap);   /* 0 = no error so far */
int tfgpu_parsequeue_close(tfgpu_parsequeue *q);
int tfgpu_parsequeue_set_release(tfgpu_parsequeue *q, tfgpu_pq_release_fn release);
void tfgpu_parsequeue_destroy(tfgpu_parsequeue *q);

/* tfgpu_bufferer = middlewares/synchronizer/bufferer (bufferer.go:16-249, buffer.go:26-56): pushes of device batches are
 * collected until TriggingCount rows, TriggingSize bytes of Values (tfgpu_dbatch_deepsizeof's total: the caller passes it),
 * TriggingInterval since the start of the last flush, or a non-row item (zero = the trigger does not apply; Close flushes
 * what is left).  One flush is in flight at a time; the next one waits for it.  A flush hands the sink ONE batch: the only
 * buffered batch itself, or — the single concat copy of buffer.Flush — tfgpu_dbatch_concat of the buffered batches on the
 * calling process's device (concat_on_device = 1; with 0 `merged` is NULL for several parts and the sink sees `parts`).
 * flush(user, merged, parts, nparts, nrows, values_size) returns the sink's error, which every buffered push's ticket then
 * carries (tfgpu_bufferer_wait; TFGPU_PQ_PENDING on a timeout, timeout_ms < 0 = wait).  The batches stay the caller's: they
 * may be freed once their ticket is answered.  async_push blocks until the collector has taken the item (the reference's
 * unbuffered input channel: backpressure).  A ticket's final answer (anything but TFGPU_PQ_PENDING) is given ONCE: the ticket is
 * forgotten with it.  A push that is answered at creation (an empty batch, a closed bufferer) returns a reserved id (bit 63
 * set) that carries its answer: it holds no state, may be waited on any number of times, or never.  src_row of a merged batch: async_push_meta states how many rows the batch's SOURCE had (meta_rows: what its
 * src_row counts in, e.g. the rows before a filter); when every buffered batch states it, part g's src_row is shifted by the
 * source rows of the parts in front of it — the sink lines its row metas up the same way; when one does not (async_push), the
 * merged batch carries no src_row at all rather than one that collides across parts (the parts keep theirs).                */
typedef struct tfgpu_bufferer tfgpu_bufferer;
typedef struct tfgpu_bufferer_stats { int64_t flush_all, flush_on_count, flush_on_size, flush_on_interval, flush_on_non_row; } tfgpu_bufferer_stats;
typedef int (*tfgpu_buf_flush_fn)(void *user, const tfgpu_dbatch *merged, const tfgpu_dbatch *const *parts, int nparts, int64_t nrows, uint64_t values_size);
int tfgpu_bufferer_create(int64_t trigging_count, uint64_t trigging_size, int64_t trigging_interval_ms, int concat_on_device, tfgpu_buf_flush_fn flush, void *user, tfgpu_bufferer **out);
int tfgpu_bufferer_async_push(tfgpu_bufferer *b, const tfgpu_dbatch *batch, int64_t nrows, uint64_t values_size, int has_non_row_item, uint64_t *ticket);
int tfgpu_bufferer_async_push_meta(tfgpu_bufferer *b, const tfgpu_dbatch *batch, int64_t nrows, uint64_t values_size, int has_non_row_item, int64_t meta_rows, uint64_t *ticket);
int tfgpu_bufferer_wait(tfgpu_bufferer *b, uint64_t ticket, int64_t timeout_ms);
int tfgpu_bufferer_get_stats(tfgpu_bufferer *b, tfgpu_bufferer_stats *out);
int tfgpu_bufferer_close(tfgpu_bufferer *b);
void tfgpu_bufferer_destroy(tfgpu_bufferer *b);

/* Parquet object → device columns (tf_parquet.hip; pkg/providers/s3/reader/registry/parquet/reader_parquet.go:137-340, which reads
 * rows through parquet-go and boxes every value, and parquet_schema_resolver.go:81-158).  `bytes` is the whole object in HOST memory
 * (the footer and the page headers are walked on the host; the object is uploaded once and every value is decoded on the device).
 *
 * tfgpu_parquet_resolve_schema = resolveSchema + the system columns (hide_system_cols = 0: `__file_name` utf8 and `__row_index`
 * uint64 in front, both keys — s3_reader.AppendSystemColsTableSchema for a schema without a key): a leaf is typed by its physical
 * type, then its logical type, then its converted type (BOOLEAN boolean, INT32 int32, INT64 int64, FLOAT float, DOUBLE double, INT96
 * utf8, BYTE_ARRAY / FIXED_LEN_BYTE_ARRAY string; DATE date, STRING / UUID / ENUM utf8, INT(…) int64 / uint64, DECIMAL double,
 * TIMESTAMP timestamp), a group is `any`; original_type = "parquet:" + the node's type string.  Free with tfgpu_schema_free.
 *
 * tfgpu_parquet_read_object = Read + constructCI + parseParquetField.  `schema` = the TableSchema given to Read: its columns are looked
 * up in the file by name (a column the file lacks is nil in every row, :256-259); NULL = every top-level field of the file.  With a
 * file_name the system columns of `schema` are the reader's own: `__file_name` = file_name, `__row_index` = the 1-based row (uint64).
 * Values are what abstract.Restore (pkg/abstract/restore.go:20-260) makes of parquet-go's physical Go values under the column's
 * DataType: bool, int32, int64, float32, float64; BYTE_ARRAY and FIXED_LEN_BYTE_ARRAY []byte → string under `string` / `utf8`;
 * int32 → int64 / uint64 and float32 → float64 when the DataType says so; INT96 → the decimal text of its 96 bits
 * (deprecated.Int96.String()); DATE → time.Time (parseLogicalDate); an INT64 under `timestamp` → ytschema.Timestamp(v).Time(), i.e.
 * MICROseconds whatever unit the file states; DECIMAL (typed double) → nil in every row (its int32 / int64 / []byte value falls to
 * Restore's default).  Data pages v1 / v2; PLAIN, dictionary, RLE (booleans), DELTA_BINARY_PACKED, DELTA_LENGTH_BYTE_ARRAY and
 * DELTA_BYTE_ARRAY (this one expanded on the host while it walks the page) encodings; UNCOMPRESSED / SNAPPY / GZIP / ZSTD / LZ4_RAW
 * chunks (a compressed object's pages are inflated on the host while it walks them).  A requested column that is a group (nested /
 * repeated: the `any` tree parquet-go builds) answers TFGPU_ERR_UNSUPPORTED naming it — the flat columns next to it are read when the
 * schema leaves the group out; BYTE_STREAM_SPLIT, BROTLI / LZO / hadoop-LZ4 likewise.  A corrupt object (a length, offset or
 * dictionary index past what is there, a page claiming more than its bytes can inflate to) is TFGPU_ERR_INVALID.
 * PARITY: the type mapping, the OriginalType strings, the Go value types and the first rows of the reference's 20 flat reader canon
 * files (tests/canon/s3/parquet/canondata → tests/golden/parquet_reader.json) are reproduced on inputs re-created from the canon's
 * values (tests/test_parquet_canon.py); its 10 files with nested columns are refused by name.  The file-format decoder itself
 * (parquet-go, not under /root/reference) is pinned to pyarrow's reading of the same files (tests/test_parquet.py).
 * tfgpu_parquet_read = tfgpu_parquet_read_object without a file name (system column names are then looked up in the file).
 * tfgpu_parquet_read_staged = tfgpu_parquet_read_object for an object somebody else's copy engine has already brought into HBM (the
 * puller of a pull / decode pipeline: the reader's upload is then not in front of its kernels): `bytes` stays the host copy the
 * footer and the page headers are walked in, `staged` (tfgpu_dbuf_alloc + tfgpu_dbuf_write, or tfgpu_dbuf_upload of a padded copy)
 * holds the same `len` bytes at offset 0 and is at least tfgpu_parquet_staging_size(bytes, len) long — the reader writes what it
 * decodes beside the object (expanded dictionary indices, DELTA_BINARY_PACKED values, INT96 texts) behind it.  The buffer stays the
 * caller's and may be freed when the call returns.  A compressed object does not use `staged`: its SNAPPY / LZ4_RAW data pages are
 * inflated on the DEVICE (pq_inflate, one wave a page: the compressed object goes up as it is; the host inflates only what its walk reads
 * at a page's front — definition levels, an index width — and dictionary / DELTA_* pages), GZIP / ZSTD pages on the host.          */
int tfgpu_parquet_resolve_schema(const void *bytes, uint64_t len, int hide_system_cols, tfgpu_schema **out);
int tfgpu_parquet_read_object(const void *bytes, uint64_t len, int mem, const tfgpu_schema *schema, const char *table_ns, const char *table_name, const char *file_name, tfgpu_dbatch **out);
int tfgpu_parquet_read(const void *bytes, uint64_t len, int mem, const tfgpu_schema *schema, const char *table_ns, const char *table_name, tfgpu_dbatch **out);
int tfgpu_parquet_staging_size(const void *bytes, uint64_t len, uint64_t *need);
int tfgpu_parquet_read_staged(const void *bytes, uint64_t len, const tfgpu_dbuf *staged, const tfgpu_schema *schema, const char *table_ns, const char *table_name, const char *file_name, tfgpu_dbatch **out);

/* Device columns → a Parquet object (tf_parquetw.hip; pkg/serializer/parquet.go:53-200 parquetBatchSerializer.Serialize + Close,
 * parquet_format.go:13-135 BuildParquetSchema / toParquetValue).  `schema` = the TableSchema: its columns become the fields of the
 * group "table" in NAME order (parquet.Group is a map), typed by primitiveTypesMap (intN → INT(N), float64 → STRING: its decimal
 * text, date → DATE, datetime / timestamp / interval → TIMESTAMP(NANOS), any → JSON), OPTIONAL unless TFGPU_COL_REQUIRED; a
 * column the batch lacks is null in every row.  `codec` = CodecFromString's names ("SNAPPY", "GZIP", "ZSTD"; anything else:
 * uncompressed).  Row groups are cut by row_group_max_rows, else by row_group_max_bytes over the PLAIN sizes (both 0: 128 MiB,
 * NewParquetBatchSerializer's default), at multiples of eight rows; one PLAIN v1 data page per column chunk.  *bytes is pinned
 * host memory of *len bytes: free it with tfgpu_host_free.  The batch's values must have the Go types Strictify gives them
 * (tfgpu_strictify first: that is NewStrictifyingBatchSerializer); a float64 VALUE (fmt's %v) is refused by name.  A nil in a
 * Required field is written as the type's zero value, as parquet-go does.  PARITY: what the object SAYS — field order, physical and
 * logical types, repetition, row count, every value — is pinned to the one Parquet object the reference holds
 * (canondata/reference.reference.TestBatchSerializer_parquet_default/result, tests/test_parquet_write.py); its BYTE layout
 * (encodings, page and footer bytes) is parquet-go's choice and stays unpinned.                                               */
int tfgpu_parquet_write(const tfgpu_dbatch *b, const tfgpu_schema *schema, const char *codec, int64_t row_group_max_rows, uint64_t row_group_max_bytes, void **bytes, uint64_t *len);

/* strictify.Strictify (pkg/abstract/changeitem/strictify/strictify.go:17-157) over a device batch — the first step of the
 * strictifying serializers (pkg/serializer/strictify.go:24-36): every column named by `schema` (NULL: the batch's own
 * TableSchema, else the columns' DataTypes) is brought to the strict Go type of its DataType; other columns are shared as they
 * are.  Device-resident conversions: a Go string / json.Number under any DataType (the CSV ingest's own cell conversions:
 * ParseInt of trimZeroDecimal, ParseBool, ParseFloat 32, castx.ToJSONNumberE, cast.StringToDate's layouts, time.ParseDuration)
 * and the integer kinds + bool under integer / bool / float / time / interval DataTypes (Go conversions, then toSignedInt /
 * toUnsignedInt's limits).  The reference fails the WHOLE call on the first value that cannot be converted and changes
 * nothing: TFGPU_ERR_INVALID with that value's row and column in *bad_row / *bad_col (rows in order, a row's columns in
 * order) and "failed to strictify the value of column [i] "name"" as the message.  A (DataType, Go kind) pair or a value form
 * that only the host decides answers TFGPU_ERR_UNSUPPORTED naming it.                                                   */
int tfgpu_strictify(const tfgpu_dbatch *in, const tfgpu_schema *schema, tfgpu_dbatch **out, int64_t *bad_row, int32_t *bad_col);

/* Pinned staging memory (hipHostMalloc) for double-buffered H2D/D2H.        */
int tfgpu_host_alloc(size_t bytes, void **out);
int tfgpu_host_free(void *p);

/* ---- transformers: abstract.Transformer -------------------------------- */
/* transformer.New(type, cfg…) — pkg/transformer/registry.go:36-47.
 * `type_name` is the YAML key ("mask_field", "filter_rows", "rename_tables",
 * "filter_columns", "skip_events", "convert_to_string", "convert_to_datetime",
 * "sharder_transformer", "replace_primary_key", "sql"); `config_json` is the
 * same JSON object the Go factory receives.
 * "sql" (pkg/transformer/registry/clickhouse/clickhouse_local.go:97-294; the
 * reference evaluates the query in an external clickhouse-local process) has
 * a device plan for the predicate + cast subset:
 *   SELECT * | col | expr [AS] alias, … FROM table [WHERE col op literal
 *   [AND …] [OR …]] with expr = integer / string literals, col ± integer,
 *   toInt8…toUInt64(expr), toString(col), toDateTime(col)
 * (transferia_amd/csrc/tf_sql.cpp).  A query outside it answers
 * TFGPU_ERR_UNSUPPORTED naming the construct — the shim keeps the stock
 * transformer for it; a malformed query answers TFGPU_ERR_CONFIG.  Apply
 * runs Collapse first like the reference; a batch holding Updates that move
 * their primary key (SplitUpdatedPKeys) answers TFGPU_ERR_UNSUPPORTED.
 * PARITY of "sql" is UNPINNED by construction: the authority is a ClickHouse
 * binary that is not here; device and oracle (oracle/ora_sql.py) restate
 * ClickHouse's documented typing and arithmetic.  That includes cityHash64:
 * both sides restate CityHash v1.0.2 from its published description and
 * agree with each other; the only value either is checked against is the
 * empty string's (= the constant k2) — a shared misreading of the 17-32 /
 * 33-64 / > 64-byte branches would not be caught here.  Row-wise expressions
 * are where the device subset ends: aggregates, joins and several
 * statements stay with the stock transformer.                                */
int tfgpu_plan_create(const char *type_name, const char *config_json, tfgpu_plan **out);
void tfgpu_plan_destroy(tfgpu_plan *plan);
const char *tfgpu_plan_type(const tfgpu_plan *plan);                /* Type()        */
int tfgpu_plan_description(const tfgpu_plan *plan, char *buf, size_t cap); /* Description() */
/* Suitable(table, schema): *out = 1/0.                                      */
int tfgpu_plan_suitable(const tfgpu_plan *plan, const char *table_ns, const char *table_name,
                        const tfgpu_schema *schema, int *out);
/* ResultSchema(schema): *out is library-owned, free with tfgpu_schema_free.  */
int tfgpu_plan_result_schema(const tfgpu_plan *plan, const tfgpu_schema *in, tfgpu_schema **out);
void tfgpu_schema_free(tfgpu_schema *s);

/* Number of registered transformer types and their names.                   */
int tfgpu_registry_count(void);
const char *tfgpu_registry_name(int i);

/* ---- batches ------------------------------------------------------------ */
/* `host->mem` TFGPU_MEM_HOST: pageable or pinned host buffers; TFGPU_MEM_DEVICE: device buffers (copied device-to-device,
 * e.g. the receive buffers of an all-to-all).  The caller's buffers are free for reuse on return.                    */
int tfgpu_batch_upload(const tfgpu_batch *host, tfgpu_dbatch **out);
/* Borrowed view with DEVICE pointers (valid until tfgpu_dbatch_free).       */
int tfgpu_dbatch_view(const tfgpu_dbatch *b, tfgpu_batch *view);
/* The batch's rows (len(items)) without touching its columns: the rows of a batch that left filter_rows / skip_events may still be a
 * SELECTION over the batch the filter read (gathered by the first call that reads columns: view, download, a serializer, any
 * transformer but mask_field) — a sink that only counts and drops (devnull) never pays for the gather.  -1 for NULL.           */
int64_t tfgpu_dbatch_nrows(const tfgpu_dbatch *b);
/* Gathers the rows of a batch that are still a selection, now (a sink that is about to read every column anyway); no-op otherwise. */
int tfgpu_dbatch_dense(const tfgpu_dbatch *b);
/* Copy into caller-allocated host buffers sized from the view — including, where the view shows them, every column's `absent` bitmap and
 * `col_order` (TFGPU_ERR_INVALID when the batch has them and host_out has no room: a row that does not list a column is not a row with a nil). */
int tfgpu_dbatch_download(const tfgpu_dbatch *b, tfgpu_batch *host_out);
void tfgpu_dbatch_free(tfgpu_dbatch *b);

/* Apply(items) for a chain of plans, the loop of transformation.do
 * (pkg/transformer/transformation.go:252-274), entirely in HBM.
 * `errs` (optional) receives up to `errs_cap` row errors; *nerrs the total.  */
int tfgpu_apply(tfgpu_plan *const *plans, int nplans, const tfgpu_dbatch *in, tfgpu_dbatch **out,
                tfgpu_row_error *errs, int64_t errs_cap, int64_t *nerrs);

/* ---- transformation.Push (pkg/transformer/transformation.go:46-282) -------------------------------------------------
 * The stage middlewares.Transformation builds from Transfer.TransformationConfigs() + ExtraTransformers
 * (pkg/middlewares/transformation.go:12-34): `transformers` in that order.  The Go shim cuts a Push into contiguous
 * same-table, same-schema runs (SplitByTableID, pkg/abstract/changeitem/utils.go:130-136, then the schema-hash cut of
 * transformation.do :236-251 — it has to anyway, a columnar batch IS such a run) and pushes each run:
 *   - the table plan = the transformers that are Suitable for (TableID, schema), each judged against the ResultSchema of
 *     its predecessors (AddTablePlan :46-85), built once per (TableID, schema) under the transformation's mutex and cached
 *     (preparePlans :93-121); tfgpu_transformation_table_plan returns it (indices into `transformers`);
 *   - the Apply loop of transformation.do (:252-274);
 *   - every TransformerError keeps its Input — the item AS THE FAILING TRANSFORMER SAW IT, i.e. after its predecessors:
 *     error_batches[g] holds the refused rows of the g-th failing transformer (index error_steps[g]) in error order, and
 *     errs (grouped the same way; row = index in the ORIGINAL run) the reason.  The shim appends the `__transform_error`
 *     column with Go's own error text and pushes them to the sink or drops them (errorChangeItems / pushErrors :173-235);
 *   - MiddlewareTransformerStats (:146-155): tfgpu_transformation_get_stats.
 * Thread-safe: runs of different tables are pushed from different goroutines (:131-135).                              */
typedef struct tfgpu_transformation tfgpu_transformation;
typedef struct tfgpu_transformation_stats {
  int64_t pushes, items_in, items_out;
  int64_t dropped;      /* sta.Dropped: incoming - transformed                */
  int64_t errors;       /* sta.Errors                                         */
  int64_t elapsed_ns;   /* sta.Elapsed, summed                                */
  int64_t plans_built;  /* AddTablePlan calls (cache misses)                  */
} tfgpu_transformation_stats;
int tfgpu_transformation_create(tfgpu_plan *const *transformers, int n, tfgpu_transformation **out); /* plans stay the caller's */
/* The chain as middlewares.Transformation builds it (pkg/middlewares/transformation.go:12-36) from the transfer's own
 * transformer.Transformers JSON (pkg/transformer/abstract.go:21-66): {"transformers":[{"<type>":{config},"transformerId":…},…],
 * "errorsOutput":{"Type":"sink"|"devnull"}}.  Every entry's type is util.Snakify of its key ("maskField" -> "mask_field"), its
 * config goes to tfgpu_plan_create as it is, the plans are owned by the transformation; `extra` = Transformation.ExtraTransformers,
 * appended after them (the caller's).  A type that has no device plan fails the call with that type's code and "unable to init:
 * <type>: …" — the shim then builds the stock chain.                                                                        */
int tfgpu_transformation_from_config(const char *transformers_json, tfgpu_plan *const *extra, int n_extra, tfgpu_transformation **out);
int tfgpu_transformation_size(const tfgpu_transformation *t);
const char *tfgpu_transformation_plan_type(const tfgpu_transformation *t, int i);
const char *tfgpu_transformation_errors_output(const tfgpu_transformation *t);   /* "sink", "devnull" or "" */
void tfgpu_transformation_destroy(tfgpu_transformation *t);
int tfgpu_transformation_table_plan(tfgpu_transformation *t, const char *table_ns, const char *table_name, const tfgpu_schema *schema,
                                    int32_t *idx, int32_t cap, int32_t *n);
/* `schema` = the run's TableSchema (NULL: the batch's own TableSchema / columns).                                       */
int tfgpu_transformation_push(tfgpu_transformation *t, const tfgpu_dbatch *in, const tfgpu_schema *schema, tfgpu_dbatch **transformed,
                              tfgpu_dbatch **error_batches, int32_t *error_steps, int32_t batches_cap, int32_t *n_error_batches,
                              tfgpu_row_error *errs, int64_t errs_cap, int64_t *nerrs);
int tfgpu_transformation_get_stats(tfgpu_transformation *t, tfgpu_transformation_stats *out);
/* The same push behind a token: it runs on one of the library's own worker threads, each bound to its own device lane
 * (stream + HBM cache + pinned ring), after everything the calling thread's lane had enqueued so far.  The parsequeue's
 * goroutines (pkg/parsequeue/parsequeue.go:57-154) submit parse results and wait for tokens in push order — no goroutine
 * has to stay locked to an OS thread.  `in` (and the plans) must stay alive until tfgpu_wait returns; a token is waited
 * for exactly once.  tfgpu_executor_start(n) sizes the pool (default 2 on first use; n < tfgpu_lane_count()).          */
typedef struct tfgpu_token tfgpu_token;
int tfgpu_executor_start(int workers);
int tfgpu_transformation_push_async(tfgpu_transformation *t, const tfgpu_dbatch *in, const tfgpu_schema *schema, tfgpu_token **token);
int tfgpu_wait(tfgpu_token *token, tfgpu_dbatch **transformed, tfgpu_dbatch **error_batches, int32_t *error_steps, int32_t batches_cap,
               int32_t *n_error_batches, tfgpu_row_error *errs, int64_t errs_cap, int64_t *nerrs);

/* abstract.Collapse (pkg/abstract/changeitem/change_item_collapse.go:48-134): the PK-keyed dedup of one batch that
 * sinks with primary keys run before writing (and clickhouse_local.go:178-210 before the `sql` transformer).  Keys are the
 * TableSchema's PrimaryKey columns (the batch must carry `schema` with TFGPU_COL_KEY flags); the key of a row is
 * json.Marshal of its key values in key-name order, taken from OldKeys for Update / Delete rows that have them
 * (change_item.go:314-358).  Inserts replace, Updates merge into the row filed under their old key (and may re-file it),
 * Deletes drop it and are kept once per key.  Output order: non-row kinds, then the surviving rows in the order of their
 * last change, then the deletes (the reference leaves those in map order; here: input order).  src_row of a surviving row
 * is the row whose Kind / OldKeys / LSN it keeps (the first of its chain); its values are the last writer's.
 * Batches with fewer than two rows, inserts only, or no key column come back unchanged.
 * Items whose ColumnNames differ (TOAST-style Updates that leave columns out) say so through tfgpu_column.absent: the
 * compareColumns merge (:7-35, :86-100) then runs on the device — a merged row lists the union of its chain's columns and
 * takes every column's value from the LAST item of the chain that lists it.  The reference appends names a later item
 * brings to the END of the merged row's ColumnNames; where that differs from batch order (an earlier item lacks a column
 * that a later item lists in FRONT of one the earlier item has) the result carries every row's own order
 * (tfgpu_batch.col_order).  An Update that merges and lists only SOME of the primary-key columns is refused by name
 * (TFGPU_ERR_UNSUPPORTED, the Go path takes the batch).  SynchronizeKind items are uploaded as TFGPU_K_SYNCHRONIZE
 * (InsertsOnly, :37-44, treats that one as an insert).  Keys are filed by 128-bit hashes of their
 * json.Marshal text and, hash for hash, compared as strings: two rows share a key exactly when the reference's map would
 * file them under one string.  TFGPU_ERR_UNSUPPORTED: NaN / Inf in a float key column, more than 2^29 rows.             */
int tfgpu_collapse(const tfgpu_dbatch *in, tfgpu_dbatch **out);

/* ChangeItem.KeysChanged (pkg/abstract/changeitem/change_item.go:237-286) for every row: changed[r] = 1 iff row r is an
 * Update and, for some PrimaryKey column of the TableSchema, reflect.DeepEqual(old value, new value) is false — old from
 * OldKeys by name (nil when the row has none), new from ColumnValues by name (nil when absent); values of different Go
 * types are never equal, NaN differs from NaN.  `changed` is a HOST array of nrows bytes.  This is the per-row test of
 * abstract.SplitUpdatedPKeys (utils.go:75-128): the shim cuts the batch at the flagged rows and turns each of them into
 * its (Delete by OldKeys, Insert) pair before Collapse (clickhouse_local.go:108, ydb/sink.go:732, yt/sink/sink.go:379). */
int tfgpu_keys_changed(const tfgpu_dbatch *in, uint8_t *changed /* [nrows] */, int64_t *nchanged /* optional */);

/* util.DeepSizeof(item.ColumnValues) of every row (pkg/util/sizeof.go:7-93) — what measurer.AsyncPush stores in
 * Size.Values for the Bufferer's byte trigger (pkg/middlewares/synchronizer/measurer.go:38-42) and what the s3 CSV reader
 * stores in Size.Read (reader_csv.go:336): 24 + per column 16 + (nil: 0 | scalar: its size | string, json.Number:
 * 16 + len | []byte: 24 + len | time.Time: 24 | `any`: the decoded value's size, numbers as json.Number, or as float64
 * with TFGPU_SIZEOF_JSON_FLOAT64).  *total = the sum over rows; per_row (HOST, nrows entries) is optional.  Text payloads
 * are not read.  One deviation: an `any` object that repeats a key is counted with every repeat (the decoded Go map keeps
 * one entry); containers nested deeper than 64 → TFGPU_ERR_UNSUPPORTED.                                              */
#define TFGPU_SIZEOF_JSON_FLOAT64 1u
int tfgpu_dbatch_deepsizeof(const tfgpu_dbatch *in, uint32_t flags, uint64_t *per_row, uint64_t *total);

/* Hash-partition, local half (BASELINE.json configs[4]: debezium stream → hash-partition → dedup → Kafka sink on 8 GPUs).
 * Rows are regrouped by part_id — what sharder_transformer wrote: PartID = itoa(CRC32_IEEE(join(SerializeToString(cols), "."))
 * % shards), sharder.go:130-145 — parts 0..nparts-1 in order, original row order kept inside a part; counts[d] = rows of
 * part d.  Every column buffer of *out is then laid out for ONE all-to-all (RCCL over xGMI) with split sizes counts[];
 * the receive side rebuilds a batch from the received device buffers with tfgpu_batch_upload(mem = TFGPU_MEM_DEVICE). */
int tfgpu_partition(const tfgpu_dbatch *in, int nparts, tfgpu_dbatch **out, int64_t *counts /* [nparts] */);

/* Hash-partition, exchange half: the path's ONE data-path collective (one process per GPU, RCCL point-to-point over xGMI).
 * The reference shards a transfer by giving each worker its own source partitions and lets sharder_transformer stamp
 * PartID on every item (sharder.go:130-145) for the sink to route by; on N GPUs "one PartID -> one consumer" becomes an
 * all-to-all of the rows tfgpu_partition grouped.
 *   tfgpu_comm_unique_id: rank 0 makes the 128-byte rendezvous id (ncclGetUniqueId); the shim hands it to the other
 *     workers over its control plane (the coordinator's key-value state, cpclient) — here: torch.distributed broadcast.
 *   tfgpu_comm_init: every rank joins (ncclCommInitRank) on the device tfgpu_init selected.  Collective.
 *   tfgpu_exchange: collective.  `in` holds this rank's rows grouped by destination rank, counts[d] rows for rank d
 *     (sum = nrows) — tfgpu_partition's output with nparts = world.  *out = the rows every rank sent to this one, source
 *     ranks in order, row order inside a source kept; recv_counts[s] = rows from rank s (optional).  All column buffers,
 *     validity, nanos, kinds, src_row, OldKeys (+ presence) travel; TableSchema / table id are taken from `in` (constant
 *     per table); part_id of *out = this rank.  Ranks must hold the same columns in the same representation: a mismatch
 *     is detected from the exchanged descriptors and fails with TFGPU_ERR_INVALID on EVERY rank, before any payload
 *     moves.  A rank whose batch has no OldKeys while others have them sends nil OldKeys named by its TableSchema keys.
 *     Runs on the calling lane's stream; RCCL itself is loaded on first use (librccl.so.1, or $TFGPU_RCCL_LIB).        */
#define TFGPU_COMM_ID_BYTES 128
typedef struct tfgpu_comm tfgpu_comm;
int tfgpu_comm_unique_id(uint8_t id[TFGPU_COMM_ID_BYTES]);
int tfgpu_comm_init(const uint8_t id[TFGPU_COMM_ID_BYTES], int rank, int world, tfgpu_comm **out);
void tfgpu_comm_destroy(tfgpu_comm *c);
int tfgpu_comm_rank(const tfgpu_comm *c);
int tfgpu_comm_world(const tfgpu_comm *c);
int tfgpu_exchange(tfgpu_comm *c, const tfgpu_dbatch *in, const int64_t *counts /* [world] */, tfgpu_dbatch **out,
                   int64_t *recv_counts /* [world], optional */);

/* ---- CSV ingest: pkg/csv.Reader + s3 CSVReader.doParse ------------------ */
typedef struct tfgpu_csv_options {
  /* pkg/csv/reader.go:43-54, defaults from NewReader (reader.go:337-350)    */
  uint8_t delimiter;        /* ','  */
  uint8_t quote_char;       /* '"'  ; 0 = quoting disabled                   */
  uint8_t escape_char;      /* '\\' ; 0 = none                               */
  uint8_t double_quote;     /* 1                                             */
  uint8_t newlines_in_value;/* 0                                             */
  /* reader_csv.go additionalReaderOptions                                   */
  uint8_t include_missing_columns;
  uint8_t strings_can_be_null;
  uint8_t quoted_strings_can_be_null;
  int32_t n_null_values;  const char *const *null_values;
  int32_t n_true_values;  const char *const *true_values;
  int32_t n_false_values; const char *const *false_values;
  int32_t n_timestamp_parsers; const char *const *timestamp_parsers; /* Go layouts */
  const char *decimal_point; /* "" or NULL = '.'                             */
  int64_t skip_rows;         /* lines dropped before parsing starts          */
  /* constructCI's system columns (reader_csv.go:275-290; names and types from
   * s3_reader.AppendSystemColsTableSchema, pkg/providers/s3/reader/util.go:210-215):
   * a schema column named "__file_name" (utf8) / "__row_index" (uint64) reads no CSV field. */
  const char *file_name;     /* value of __file_name; NULL = ""                            */
  uint64_t row_number_base;  /* rowsCounter of the first parsed line (reader_csv.go:98: 1); every
                              * line read advances it, failed or not (:217-220)            */
  uint8_t hide_system_cols;  /* config.hideSystemCols: both system columns are nil         */
  /* csv.Reader.Encoding (reader.go:157-183): NULL = UTF-8 input.  Otherwise the 256 code points
   * the x/text charmap decoder of that name yields for bytes 0..255 (the shim reads them off
   * enc.NewDecoder() once); every line is decoded after it was cut at the raw '\n' byte.    */
  const uint32_t *encoding_table;
} tfgpu_csv_options;
void tfgpu_csv_options_default(tfgpu_csv_options *o);

/* Parse `len` bytes of CSV into a device batch typed by `schema` (column i
 * reads CSV field atoi(schema.cols[i].path)), then Strictify.  *consumed is
 * the offset of the first unconsumed byte (a trailing line without '\n' is
 * left for the next chunk, reader.go:158-168).
 * Text columns of the result are LATE-MATERIALISED: like the Go substrings that alias the chunk they were cut from,
 * a cell is (length, position in the CSV text) until a consumer needs packed bytes; row filters then pack only the
 * kept cells, straight from the text.  The library keeps the text alive itself: a host chunk is staged in HBM anyway,
 * a tfgpu_dbuf is shared (freeing the handle is safe; tfgpu_dbuf_write on a buffer a batch still reads goes to a
 * private copy).  Any OTHER device pointer (mem = TFGPU_MEM_DEVICE, not a tfgpu_dbuf base) is only trusted for the
 * duration of the call: the columns are packed before it returns.  TFGPU_CSV_EAGER=1 forces that everywhere.      */
int tfgpu_csv_parse(const tfgpu_csv_options *opts, const tfgpu_schema *schema, const void *bytes,
                    uint64_t len, int mem, tfgpu_dbatch **out, uint64_t *consumed,
                    tfgpu_row_error *errs, int64_t errs_cap, int64_t *nerrs);

/* csv.Splitter (pkg/csv/splitter.go:37-85): where ConsumeRow cuts a byte stream into CSV entries — at the '\n's outside
 * double quotes (its three states reduce to the parity of the '"' seen so far).  *row_ends = uint32 offsets one past the
 * '\n' of every complete entry, in order (a tfgpu_dbuf of *nrows words); what follows the last one is the io.EOF remainder. */
int tfgpu_csv_split_rows(const void *bytes, uint64_t len, int mem, tfgpu_dbuf **row_ends, int64_t *nrows);

/* ---- JSON / TSKV ingest: parsers/generic GenericParser{Format:"json" | "tskv"} ----------- */
/* generic.AuxParserOpts (pkg/parsers/generic/generic_parser.go:40-77) as the "json" parser
 * registry fills them (pkg/parsers/registry/json/parser_json.go:62-86).  Options whose value
 * forms are not device-resident are rejected with TFGPU_ERR_UNSUPPORTED at call time
 * (unescape_string_values, unpack_bytes_base64, TimeField, TableSplitter).  A nested ColSchema.Path
 * ("EventValue.LogInfo", "a/b/c": lookupComplex + parseJSON, pkg/parsers/generic/lookup.go:10-59) is
 * walked on the device for string / null targets, the retries of parseJSON included; lines whose
 * target is a number, bool or container, or whose member names are not plain ASCII, come back as
 * TFGPU_ROW_HOST_FALLBACK; together with AddRest it is TFGPU_ERR_UNSUPPORTED.                        */
typedef struct tfgpu_json_options {
  uint8_t add_rest;               /* AddRest: `_rest` column (any)                           */
  uint8_t add_dedupe_keys;        /* AddDedupeKeys: _timestamp,_partition,_offset,_idx       */
  uint8_t null_keys_allowed;      /* NullKeysAllowed                                         */
  uint8_t use_numbers_in_any;     /* UseNumbersInAny                                         */
  uint8_t unescape_string_values; /* UnescapeStringValues                                    */
  uint8_t unpack_bytes_base64;    /* UnpackBytesBase64                                       */
  uint8_t ignore_column_paths;    /* IgnoreColumnPaths                                       */
  uint8_t mark_dedupe_keys_as_system; /* MarkDedupeKeysAsSystem (json.lb: SkipSystemKeys)     */
  const char *topic;              /* AuxOpts.Topic → GenericParser.name → ChangeItem.Table   */
  const char *partition;          /* abstract.Partition.String() of the batch (PartID)       */
  uint8_t format;                 /* GenericParserConfig.Format: TFGPU_JFMT_JSON (0) or TFGPU_JFMT_TSKV — tab-separated
                                     key=value fields, every value a Go string (generic_parser.go:732-746); with
                                     unescape_string_values: tryToUnescapeTSKV (:643-670) on text columns         */
} tfgpu_json_options;
enum { TFGPU_JFMT_JSON = 0, TFGPU_JFMT_TSKV = 1 };

/* parsers.MessageBatch (pkg/parsers/abstract.go): Messages[i].Value concatenated in `bytes`.   */
typedef struct tfgpu_messages {
  int64_t nmsg;
  const uint64_t *start;          /* [nmsg+1] byte offset of each Value; start[nmsg] == len   */
  const uint64_t *offset;         /* Message.Offset → LSN, `_offset`                          */
  const int64_t *write_time_ns;   /* Message.WriteTime.UnixNano() → CommitTime, `_timestamp`  */
} tfgpu_messages;

/* GenericParser.DoBatch (generic_parser.go:406-438) for Format "json": every message is split
 * into lines (bufio.ScanLines), each non-empty line is one JSON object → one row typed by
 * `fields` (+ the aux columns the options add, generic_parser.go:99-154).  Lines are numbered
 * 0.. over the whole batch counting non-empty lines only ("ordinal").  Outputs:
 *   - *out: parsed rows; src_row[r] = ordinal of the line, part_id[r] = index of its message
 *     (the shim re-attaches LSN / CommitTime / QueueMessageMeta from it);
 *   - errs: one entry per line that the reference turns into an `_unparsed` row
 *     (row = ordinal, step = message index, code = TFGPU_ROW_JSON_SYNTAX / PARSE_VAL / NIL_KEY)
 *     or that must be re-parsed by the stock Go code (TFGPU_ROW_HOST_FALLBACK);
 *   - lines whose top-level value is not an object, or is `{}`, yield nothing
 *     (generic_parser.go:536: len(item) > 0), exactly like the reference.
 * `msgs` may be NULL: one message spanning the whole buffer, Offset 0, WriteTime 0.          */
int tfgpu_json_parse(const tfgpu_json_options *opts, const tfgpu_schema *fields, const void *bytes, uint64_t len,
                     int mem, const tfgpu_messages *msgs, tfgpu_dbatch **out, tfgpu_row_error *errs,
                     int64_t errs_cap, int64_t *nerrs);
/* GenericParser.ResultSchema(): fields + aux columns; free with tfgpu_schema_free.            */
int tfgpu_json_result_schema(const tfgpu_json_options *opts, const tfgpu_schema *fields, tfgpu_schema **out);

/* ---- Confluent Schema Registry ingest, JSON schemas (configs[2]) ------------------------------ */
/* ConfluentSrImpl.DoBatch / Do / DoBuf / DoOne (pkg/parsers/registry/confluentschemaregistry/engine/parser.go:108-152):
 * every Kafka message is a run of FRAMES  0x00 | schema id (BE uint32) | payload,  the payload of a JSON-schema frame
 * ending at the next 0x00 byte (format_json.go:34-38).  tfgpu_sr_frames lists them — the shim needs the schema ids to
 * ask the registry (network, stays in Go).  A short message or a wrong magic byte ends its message with an error frame. */
typedef struct tfgpu_sr_frame {
  int64_t msg;        /* index of the Kafka message                                                        */
  uint64_t start;     /* payload offset in the buffer (error frames: where the frame starts)               */
  uint32_t len;       /* payload bytes (error frames: the rest of the message)                             */
  uint32_t schema_id;
  int32_t code;       /* TFGPU_ROW_OK / TFGPU_ROW_SR_SHORT / TFGPU_ROW_SR_MAGIC                            */
  int32_t index;      /* position of the frame inside its message (QueueMessageMeta.Index)                 */
} tfgpu_sr_frame;
int tfgpu_sr_frames(const void *bytes, uint64_t len, int mem, const tfgpu_messages *msgs, tfgpu_sr_frame *frames,
                    int64_t cap, int64_t *nframes);
struct tfgpu_sr_schema; struct tfgpu_sr_property;
int tfgpu_sr_compile_schema(const char *schema_text, uint64_t len, const char *policy, const char *manual_table_name, struct tfgpu_sr_schema **out);
int tfgpu_sr_schema_info(const struct tfgpu_sr_schema *s, const struct tfgpu_sr_property **props, int32_t *nprops, const char **table_ns, const char **table_name, const char **title);
void tfgpu_sr_schema_free(struct tfgpu_sr_schema *s);

/* One property of the JSON schema as jsonPropertyToJSONSchemaRow resolves it (utils_json.go:71-95, types_json.go:25-32):
 * the `type` (through oneOf), and whether it is required (listed in "required" and no oneOf null).                  */
enum { TFGPU_SRT_BOOLEAN = 1, TFGPU_SRT_INTEGER = 2, TFGPU_SRT_NUMBER = 3, TFGPU_SRT_STRING = 4, TFGPU_SRT_ANY = 5 };
typedef struct tfgpu_sr_property { const char *name; int32_t json_type; int32_t required; } tfgpu_sr_property;
/* The per-schema set-up behind the ABI (tf_dbzrecv.cpp): unmarshal the JSON schema text the registry returned, resolve every
 * property (utils_json.go:15-21, 71-95; types_json.go:25-32), derive the table id (BuildJSONTableID, table_name_policy.go:73-92;
 * policy "debezium_style" — the default — or "title"; a non-empty manual_table_name wins).  tfgpu_sr_schema_info: the properties
 * in util.MapKeysInOrder order and the table id — what tfgpu_sr_json_options takes.  TFGPU_ERR_CONFIG carries the reference's
 * own error texts (a schema whose type is not "object", a title that does not split, a property type without a column type).   */
typedef struct tfgpu_sr_schema tfgpu_sr_schema;
typedef struct tfgpu_sr_json_options {
  uint32_t schema_id;           /* frames carrying another id are left to another call                      */
  int32_t nprops;
  const tfgpu_sr_property *props; /* util.MapKeysInOrder(Properties): sorted by name, unique               */
  const char *table_ns;         /* BuildJSONTableID(tableNamePolicy, title) (table_name_policy.go:73-92)     */
  const char *table_name;
  int32_t is_generate_updates;  /* isGenerateUpdates (format_json.go:44-47): every item is an Update and lists only the optional fields its
                                   payload holds (utils_json.go:57-63) — the others are ABSENT cells (tfgpu_column.absent) */
  int32_t report_frame_errors;  /* also report the SR_SHORT / SR_MAGIC frames (set it in one call per batch) */
} tfgpu_sr_json_options;
/* makeChangeItemsFromMessageWithJSON + processPayload + convertTypes (format_json.go:15-67, utils_json.go:27-128) for
 * every frame of `schema_id`: the payload is decoded like encoding/json's Decoder with UseNumber into a map, and every
 * property becomes one column — boolean → bool, integer → int64 (json.Number.Int64), number → json.Number (its text),
 * string → string, anything else → `any` (json.Marshal text of the decoded value).  Rows: one per good frame, Kind
 * insert, src_row = ordinal of the frame over the whole batch, part_id = index of its message.  errs: one per frame the
 * reference turns into an `_unparsed` item (row = frame ordinal, step = message index) — after which the rest of that
 * message is dropped, as DoBuf does.  An `any` value whose objects hold their keys in another order than json.Marshal's
 * (ascending, the last duplicate wins) is re-emitted sorted on the device; TFGPU_ROW_HOST_FALLBACK is left for such a
 * value nested deeper than 16 containers and for payloads nested deeper than 128.  With several schema ids in one
 * message the shim applies the "first error ends the message" rule across its calls.                                */
int tfgpu_sr_json_parse(const tfgpu_sr_json_options *opts, const void *bytes, uint64_t len, int mem, const tfgpu_messages *msgs,
                        tfgpu_dbatch **out, tfgpu_row_error *errs, int64_t errs_cap, int64_t *nerrs);

/* ---- Confluent Schema Registry ingest, PROTOBUF schemas ("then JSON/Protobuf") -------------------------------------
 * makeChangeItemsFromMessageWithProtobuf (pkg/parsers/registry/confluentschemaregistry/engine/format_protobuf.go:16-90): a Kafka
 * message is  0x00 | schema id (BE uint32) | message indexes | protobuf bytes — ONE item per Kafka message (doWithSchema consumes
 * the whole rest, parser.go:46-48).  The message is unmarshalled by the descriptor compiled from the registry's .proto text
 * (mdBuilder.toMD, md_builder.go:26-70) and unpacked field by field (unpackProtobufDynamicMessage, utils_protobuf.go:87-112):
 * one column per field of the message in declaration order, typed by protoSchemaTypes (types_protobuf.go:16-35: int32 / sint32 /
 * sfixed32 -> int32, …, string -> utf8, bytes -> string, enum -> utf8 holding the int32 number, message -> any); an absent field is
 * its proto3 zero value, an absent message field nil; kind Insert; table id from BuildProtobufTableID (table_name_policy.go:51-71).
 *
 * tfgpu_sr_compile_proto (tf_protoschema.cpp): the descriptor for what the device decodes — proto3 messages whose fields are singular
 * scalars / enums, repeated scalars / enums (`any` columns: the array of their elements, packed or not, `[]` when absent), or singular
 * messages of singular scalar / enum fields (`any` columns: the map of ALL their fields, keys sorted; confluent.type.Decimal is built in).  tfgpu_pb_schema_info: code = TFGPU_ROW_OK and the fields, TFGPU_ROW_SR_PROTO (the text does
 * not compile / the record name does not split: every message of the schema is `_unparsed`) or TFGPU_ROW_HOST_FALLBACK with `why`
 * (a oneof inside a NESTED message, maps with other than string keys or with message values, proto2, deeper nesting, other imports: the stock code; a top-level oneof's members are columns, tfgpu_pb_field.oneof).  policy: "debezium_style" (default) |
 * "message_name"; message_name: "" = the first message of the file (getRecordName).
 * tfgpu_sr_proto_parse: every Kafka message whose prefix carries schema_id.  *out: one row per good message (src_row = part_id =
 * message index); errs: row = step = message index, code = TFGPU_ROW_SR_PROTO (the bytes do not unmarshal), TFGPU_ROW_SR_SHORT /
 * TFGPU_ROW_SR_MAGIC (only with report_frame_errors), TFGPU_ROW_HOST_FALLBACK (message indexes other than the single 0 byte, a known
 * field met with another wire type, a message field met twice, groups, a NaN / Inf inside an `any` value — it has no JSON text).  With a schema whose code
 * is not TFGPU_ROW_OK every message of the id gets that code.  Parity: the .proto compiler and the dynamic message are dependencies
 * of the reference; the oracle restates the published language subset and wire format and is pinned to the reference's two PROTOBUF
 * test vectors (parser_test.go TestClient: schemas 5 and 6, test_protobuf_{0,1}.bin, canon).                                     */
enum { TFGPU_PB_DOUBLE = 1, TFGPU_PB_FLOAT, TFGPU_PB_INT64, TFGPU_PB_UINT64, TFGPU_PB_INT32, TFGPU_PB_FIXED64, TFGPU_PB_FIXED32, TFGPU_PB_BOOL, TFGPU_PB_STRING,
       TFGPU_PB_BYTES, TFGPU_PB_UINT32, TFGPU_PB_SFIXED32, TFGPU_PB_SFIXED64, TFGPU_PB_SINT32, TFGPU_PB_SINT64, TFGPU_PB_ENUM, TFGPU_PB_MESSAGE };
typedef struct tfgpu_pb_member { const char *name; int32_t number; int32_t ptype; } tfgpu_pb_member;
typedef struct tfgpu_pb_field { const char *name; int32_t number; int32_t ptype; int32_t nmembers; const tfgpu_pb_member *members; /* of a MESSAGE field, sorted by name */
                                int32_t repeated; /* 1: a repeated scalar / enum / one-level message field: an `any` column, the JSON array of its elements (a message element: its map);
                                                     2: a map<string, V> field (V scalar / enum; members = {key, value}): an `any` column, {"key":value,...} with the keys in byte order */
                                int32_t oneof;    /* k > 0: a member of the message's k-th oneof — a column like any other; on the wire a member CLEARS the group's other members
                                                     (the last one met is the one set; the others read their zero value / nil) */ } tfgpu_pb_field;
typedef struct tfgpu_pb_schema tfgpu_pb_schema;
int tfgpu_sr_compile_proto(const char *schema_text, uint64_t len, const char *policy, const char *manual_table_name, const char *message_name, tfgpu_pb_schema **out);
int tfgpu_pb_schema_info(const tfgpu_pb_schema *s, int32_t *code, const tfgpu_pb_field **fields, int32_t *nfields, const char **table_ns, const char **table_name, const char **record, const char **why);
void tfgpu_pb_schema_free(tfgpu_pb_schema *s);
int tfgpu_sr_proto_parse(const tfgpu_pb_schema *s, uint32_t schema_id, int32_t report_frame_errors, const void *bytes, uint64_t len, int mem, const tfgpu_messages *msgs,
                         tfgpu_dbatch **out, tfgpu_row_error *errs, int64_t errs_cap, int64_t *nerrs);

/* ---- serialize ---------------------------------------------------------- */
enum {
  TFGPU_FMT_CH_JSON_EACH_ROW = 1, /* httpuploader.MarshalCItoJSON             */
  TFGPU_FMT_JSON = 2,             /* pkg/serializer/json.go (sorted keys)     */
  TFGPU_FMT_CSV = 3,              /* pkg/serializer/csv.go                    */
  TFGPU_FMT_RAW = 4               /* pkg/serializer/raw.go:24-63: a mirror item's `data` bytes (+ '\n' with AddClosingNewLine; the batch
                                     serializer joins with '\n' otherwise, batch_factory.go:36-47); any other item: an error */
};
/* Target column of the ClickHouse sink — the fields of columntypes.TypeDescription
 * (pkg/providers/clickhouse/columntypes) that marshalValue / marshalTime read
 * (httpuploader/marshal.go:65-80, 140-301).                                 */
enum {
  TFGPU_CH_STRING = 1u,     /* colType.IsString                               */
  TFGPU_CH_DATE = 2u,       /* colType.IsDate                                 */
  TFGPU_CH_DATETIME64 = 4u, /* colType.IsDateTime64 (+ precision)             */
  TFGPU_CH_DECIMAL = 8u,    /* colType.IsDecimal                              */
  TFGPU_CH_ARRAY = 16u      /* colType.IsArray                                */
};
typedef struct tfgpu_serialize_options {
  int32_t add_closing_newline; /* JSONSerializerConfig.AddClosingNewLine (json.go:13-17);
                                  batch separator "\n" when 0 (batch_factory.go:36-39)    */
  int32_t any_as_string;       /* AnyAsString                                              */
  int32_t ncols;               /* CH: entries below, one per batch column; 0 = derive the
                                  target type from the DataType (clickhouse/typesystem.md) */
  const uint32_t *ch_flags;    /* TFGPU_CH_*                                               */
  const uint8_t *ch_precision; /* DateTime64 precision 0..9                                */
} tfgpu_serialize_options;
/* batchSerializer.Serialize (pkg/serializer/batch.go:73-117) / one MarshalCItoJSON per row
 * (marshal.go:82-125).  tfgpu_serialize == tfgpu_serialize_ex with NULL options.          */
int tfgpu_serialize(int format, const tfgpu_dbatch *b, tfgpu_dbuf **out);
int tfgpu_serialize_ex(int format, const tfgpu_dbatch *b, const tfgpu_serialize_options *opts, tfgpu_dbuf **out);
/* batchSerializer as a component (pkg/serializer/batch.go:20-209; NewBatchSerializer, batch_factory.go:31-63): the
 * BatchSerializerConfig, Serialize's parts of `threshold` items joined by the separator with ONE trailing separator trimmed
 * (:84-116; the undivided path :75-82 trims nothing), SerializeAndWrite's ordered Write calls (:119-209).  The device
 * serializes the batch at once — the joined parts are the batch's text — so the call adds the trim (for_writer == 0) or
 * reports where each Write of the reference ends (for_writer != 0: part_ends[0 .. *nparts), byte offsets into *out; the shim
 * replays writer.Write per part).  concurrency 0 = runtime.GOMAXPROCS(0), which the caller states in `gomaxprocs`;
 * threshold 0 = DefaultBatchSerializerThreshold (25 000); disable_concurrency = one part, no strictifying wrapper (the
 * StrictifyingSerializer of the concurrent form is tfgpu_strictify, called first by the shim).  Formats: RAW, JSON, CSV.   */
typedef struct tfgpu_batch_serializer_config {
  int32_t concurrency, threshold, disable_concurrency, gomaxprocs;
} tfgpu_batch_serializer_config;
int tfgpu_serialize_batch(int format, const tfgpu_dbatch *b, const tfgpu_serialize_options *opts, const tfgpu_batch_serializer_config *cfg,
                          int for_writer, tfgpu_dbuf **out, uint64_t *part_ends, int64_t part_cap, int64_t *nparts);
int tfgpu_dbuf_size(const tfgpu_dbuf *b, uint64_t *out);
void *tfgpu_dbuf_ptr(const tfgpu_dbuf *b); /* device pointer */
int tfgpu_dbuf_download(const tfgpu_dbuf *b, void *host, uint64_t cap);
void tfgpu_dbuf_free(tfgpu_dbuf *b);
/* Upload raw bytes (e.g. a CSV chunk) so a parse can start HBM-resident.    */
int tfgpu_dbuf_upload(const void *host, uint64_t len, tfgpu_dbuf **out);
/* Allocate `len` bytes in HBM and fill them piecewise from a (pinned) staging
 * buffer: the H2D half of the double-buffered pull loop (parsequeue.go:57-154). */
int tfgpu_dbuf_alloc(uint64_t len, tfgpu_dbuf **out);
int tfgpu_dbuf_write(tfgpu_dbuf *b, uint64_t offset, const void *host, uint64_t len);

/* ---- Debezium ingest, inline schemas (the source of BASELINE.json configs[4]) ------------------------------- */
/* DebeziumImpl.DoBatch with no schema registry (pkg/parsers/registry/debezium/engine/parser.go:33-130): every Kafka
 * message is ONE event {"schema": …, "payload": …} and becomes one ChangeItem or one `_unparsed` item.  Two steps, because
 * the Kafka Connect schema is data: the shim compiles it (host, once per distinct schema — the reference caches by the
 * schema's hash too, receiver.go:60-96) and the device does everything per message.
 *
 * tfgpu_debezium_unpack — IncludeSchema.Unpack (pkg/debezium/unpacker/include_schema.go:13-25) for every message: the
 * whole message is validated as one JSON value and the raw "schema" / "payload" members are located (exact key; the last
 * duplicate wins).  frames[m] (HOST, nmsg entries): their spans, a 128-bit hash of the schema bytes (equal bytes ⇔ equal
 * hash for grouping; FNV-style, not cryptographic) and code = TFGPU_ROW_OK / TFGPU_ROW_DBZ_UNPACK / TFGPU_ROW_HOST_FALLBACK
 * (keys that match only by case folding or carry escapes, nesting deeper than 128).  A message without "payload" keeps
 * payload_len = 0 (→ TFGPU_ROW_DBZ_PAYLOAD in the parse), without "schema" schema_len = 0.                              */
typedef struct tfgpu_dbz_frame {
  uint64_t schema_start;  uint64_t payload_start;
  uint32_t schema_len;    uint32_t payload_len;
  uint64_t schema_hash[2];
  int32_t code;           int32_t reserved;
} tfgpu_dbz_frame;
int tfgpu_debezium_unpack(const void *bytes, uint64_t len, int mem, const tfgpu_messages *msgs, tfgpu_dbz_frame *frames);
/* The same with the shim's schema cache (receiver.go:61-66 caches by the schema's hash) handed in: `known` is the head of a
 * message of an EARLIER batch up to its payload value — `{"schema":{…},"payload":` — as tfgpu_debezium_unpack framed it
 * (bytes [message start, payload_start); schema_off = schema_start - message start).  Messages that begin with exactly these
 * bytes inherit its schema span and hash and only their payload value and closing brace are walked; every other message takes
 * the full walk.  The result equals tfgpu_debezium_unpack's; what is saved is the one serial 12 KB walk per batch.          */
typedef struct tfgpu_dbz_prefix {
  const void *bytes;        /* HOST */
  uint32_t len;
  uint32_t schema_off, schema_len;
  uint32_t reserved;
  uint64_t schema_hash[2];
} tfgpu_dbz_prefix;
int tfgpu_debezium_unpack_cached(const void *bytes, uint64_t len, int mem, const tfgpu_messages *msgs, const tfgpu_dbz_prefix *known,
                                 tfgpu_dbz_frame *frames);

/* The receiver of one field of the before / after struct, as receiveFieldColSchema resolves it with an empty original type
 * (pkg/debezium/receiver_engine.go:108-146, common/field_receiver_default.go:14-31): Kafka type → Go value → YT type.   */
enum {
  TFGPU_DBZ_BOOLEAN = 1, /* boolean → bool                                                "boolean" */
  TFGPU_DBZ_INT8,        /* int8    → int8(json.Number.Int64())                            "int8"    */
  TFGPU_DBZ_INT16,       /* int16   → int16(…)                                             "int16"   */
  TFGPU_DBZ_INT32,       /* int32   → int32(…)                                             "int32"   */
  TFGPU_DBZ_INT64,       /* int64   → int64                                                "int64"   */
  TFGPU_DBZ_FLOAT64,     /* float / double → json.Number.Float64()                         "double"  */
  TFGPU_DBZ_STRING,      /* string (any logical name) → the string; a number → its text    "utf8"    */
  TFGPU_DBZ_BYTES,       /* bytes → base64.StdEncoding.DecodeString                        "string"  */
  TFGPU_DBZ_DECIMAL,     /* bytes, org.apache.kafka.connect.data.Decimal → Base64ToNumeric(value, parameters.scale)  "utf8" */
  TFGPU_DBZ_POINT,       /* struct io.debezium.data.geometry.Point → "(x,y)"               "utf8"    */
  TFGPU_DBZ_VSD,         /* struct io.debezium.data.VariableScaleDecimal → json.Number     "double"  */
  TFGPU_DBZ_HOST         /* arrays, __dt_original_type_info: every message of the schema → TFGPU_ROW_HOST_FALLBACK */
};
typedef struct tfgpu_dbz_field { const char *name; int32_t op; int32_t optional; int32_t scale; int32_t reserved; } tfgpu_dbz_field;
typedef struct tfgpu_dbz_options {
  uint64_t schema_hash[2];       /* frames with another hash are left to another call                                  */
  int32_t nfields;               /* Schema.FindAfterSchema().Fields, in order; the before struct must list the same
                                    fields (the shim checks; otherwise it keeps the schema on the host)                 */
  const tfgpu_dbz_field *fields;
  int32_t schema_code;           /* TFGPU_ROW_OK; or what receiveSchema decided for the whole schema (TFGPU_ROW_DBZ_SCHEMA:
                                    it does not unmarshal / a Kafka type without receiver; TFGPU_ROW_HOST_FALLBACK) — the
                                    payload and op checks still run first, as in Receiver.receive, and no row is produced */
  int32_t reserved;
} tfgpu_dbz_options;
/* Per produced row, the ChangeItem members that are not columns (receiver.go:187-209).                                  */
typedef struct tfgpu_dbz_row {
  int64_t msg;            /* index of the Kafka message (= src_row of the row)                                          */
  uint64_t lsn;           /* payload.source.lsn                                                                         */
  uint64_t commit_time;   /* payload.source.ts_ms * 1 000 000                                                           */
  uint32_t id;            /* payload.source.txId                                                                        */
  uint8_t names_form;     /* 1 for Delete: ColumnNames / ColumnValues stay nil (tfgpu_row_meta.names_form)              */
  uint8_t reserved[3];
} tfgpu_dbz_row;
/* Receiver.receive for every frame of opts->schema_hash (pkg/debezium/receiver.go:150-232): the payload is decoded like
 * Decoder(UseNumber).Decode(&Payload); op → kind (c, r: Insert; u: Update; d: Delete); values from `before` for Delete,
 * else `after`; every schema field must be present and is converted by its receiver (null → nil).  *out: one row per good
 * message in message order — columns = the schema fields (TableSchema: key = !optional, table_schema / table_name =
 * source.schema / source.table), kinds, src_row = message index; Update / Delete rows carry OldKeys = the key fields'
 * values (of `after` for Update — what Receiver.add stores — of `before` for Delete); a Delete row's cells are nil and
 * rows[r].names_form = 1.  rows (HOST, rows_cap entries) receives ID / LSN / CommitTime.  errs: one entry per message of
 * this schema that the reference turns into an `_unparsed` item (row = step = message index, code = TFGPU_ROW_DBZ_*) or
 * that the stock code must redo (TFGPU_ROW_HOST_FALLBACK: `__debezium_unavailable_value`, decimals wider than 64 bytes,
 * values on which the reference panics, rows of another table than the first good row's, a payload / source key that
 * repeats — encoding/json decodes every occurrence into the same struct field and merges maps — or binds only by case
 * folding).                              */
int tfgpu_debezium_parse(const tfgpu_dbz_options *opts, const void *bytes, uint64_t len, int mem, const tfgpu_messages *msgs,
                         const tfgpu_dbz_frame *frames, tfgpu_dbatch **out, tfgpu_dbz_row *rows, int64_t rows_cap,
                         tfgpu_row_error *errs, int64_t errs_cap, int64_t *nerrs);

/* The host half of the receiver (tf_dbzrecv.cpp) — what the reference does once per distinct schema, and DoBatch's loop.
 *
 * tfgpu_debezium_compile_schema — Receiver.receiveSchema (pkg/debezium/receiver.go:60-96) for one schema's bytes: UnmarshalSchema
 * with encoding/json's struct binding (debezium_schema.go:12-29; the last duplicate of a key wins, a key that binds only by case
 * folding -> host), the `before` / `after` structs, every field's receiver with an empty original type (receiveFieldColSchema,
 * receiver_engine.go:108-146; TypeToDefault and the Point / VariableScaleDecimal / Decimal matchers,
 * common/field_receiver_default.go:14-31, 258-355).  tfgpu_dbz_schema_info: code = TFGPU_ROW_OK and the `after` struct's fields
 * in order (the tfgpu_dbz_options.fields of tfgpu_debezium_parse), or what the whole schema is: TFGPU_ROW_DBZ_SCHEMA (it does
 * not unmarshal, a Kafka type without receiver) / TFGPU_ROW_HOST_FALLBACK (before != after, a repeated field name, a nil
 * before / after struct, folded keys); `why` says which.
 *
 * tfgpu_dbz_receiver — DebeziumImpl over message batches (parser.go:120-130) with the reference's schema cache (receiver.go:61-66,
 * keyed by the device's hash of the schema bytes) and the head of the opening message kept for tfgpu_debezium_unpack_cached.
 * tfgpu_dbz_receive: unpack, group the messages by schema, compile the schemas seen for the first time, parse every group;
 * *ngroups tables came out (in order of first appearance), msg_codes[m] (HOST, nmsg entries, optional) = TFGPU_ROW_OK or why
 * message m becomes an `_unparsed` item / goes to the stock code.  host_copy: the same bytes in host memory when `bytes` is a
 * device buffer (a new schema's text is read there instead of being copied back), or NULL.  tfgpu_dbz_receive_group: table g's
 * rows — the device batch (src_row = message index; the caller's from then on), per-row ID / LSN / CommitTime / names_form,
 * and the schema's fields (TableSchema: PrimaryKey = !optional, TableSchema / TableName = the batch's table id); rows and fields
 * stay valid until the next tfgpu_dbz_receive on this receiver.                                                             */
typedef struct tfgpu_dbz_schema tfgpu_dbz_schema;
int tfgpu_debezium_compile_schema(const void *schema_bytes, uint64_t len, tfgpu_dbz_schema **out);
int tfgpu_dbz_schema_info(const tfgpu_dbz_schema *s, int32_t *code, const tfgpu_dbz_field **fields, int32_t *nfields, const char **why);
void tfgpu_dbz_schema_free(tfgpu_dbz_schema *s);
typedef struct tfgpu_dbz_receiver tfgpu_dbz_receiver;
int tfgpu_dbz_receiver_create(tfgpu_dbz_receiver **out);
void tfgpu_dbz_receiver_destroy(tfgpu_dbz_receiver *r);
int tfgpu_dbz_receiver_known(const tfgpu_dbz_receiver *r, tfgpu_dbz_prefix *known);  /* the head it keeps (len 0: none yet); bytes stay the receiver's */
int tfgpu_dbz_receive(tfgpu_dbz_receiver *r, const void *bytes, uint64_t len, int mem, const void *host_copy, const tfgpu_messages *msgs, int32_t *ngroups, int32_t *msg_codes);
int tfgpu_dbz_receive_group(tfgpu_dbz_receiver *r, int32_t g, tfgpu_dbatch **batch, const tfgpu_dbz_row **rows, int64_t *nrows, const tfgpu_dbz_field **fields, int32_t *nfields);
/* table g's ID / LSN / CommitTime / names_form by MESSAGE index — the arrays a tfgpu_row_meta of the whole message batch takes
 * (nmsg entries each, zeroed first; NULL: not wanted): the rows' src_row is the message index, so this is their row meta.      */
int tfgpu_dbz_receive_group_meta(tfgpu_dbz_receiver *r, int32_t g, int64_t nmsg, uint32_t *ids, uint64_t *lsns, uint64_t *commit_times, uint8_t *names_form);

/* ---- Debezium events framed by a schema registry (NewDebeziumImpl with a registry client; the f1 remainder of SURVEY §8) ---------
 * DebeziumImpl.DoOne (pkg/parsers/registry/debezium/engine/parser.go:33-57) cuts a Kafka message into EVENTS  0x00 | schema id
 * (BE uint32) | payload up to the next 0x00 byte — the cut tfgpu_sr_frames makes; SchemaRegistry.Unpack
 * (pkg/debezium/unpacker/schema_registry.go:18-34) fetches the schema by id (network: stays with the shim) and
 * Receiver.convertSchemaFormat (receiver.go:118-139) turns the registry's ConfluentJSONSchema into the Kafka Connect form the
 * receiver reads (ToKafkaJSONSchema, pkg/schemaregistry/format/json_schema_format.go:120-164).
 *
 * tfgpu_debezium_compile_registry_schema: convertSchemaFormat + receiveSchema for one registry schema text — encoding/json's
 * binding of ConfluentJSONSchema (every field's JSON type checked), oneOf → optional, properties ordered by connect.index,
 * confluentTypeToKafka; then what tfgpu_debezium_compile_schema does.  Same result object (tfgpu_dbz_schema_info).
 * tfgpu_dbz_receiver_add_registry_schema: the same, kept by the receiver under `schema_id`.
 * tfgpu_debezium_registry_frames: the first half of UnmarshalPayload for every event of tfgpu_sr_frames — the span of the ONE value
 * json.Decoder reads behind the prefix (frames[e].payload_*; code = TFGPU_ROW_DBZ_PAYLOAD when it is not JSON,
 * TFGPU_ROW_SR_MAGIC for a first byte that is not 0 — DoOne looks at it before the length — TFGPU_ROW_HOST_FALLBACK for a zero-led tail
 * shorter than the prefix: the reference's buf[5:] panics);
 * frames[e].schema_hash = {schema id, TFGPU_DBZ_REGISTRY_HASH}.  event_msgs: one slot per event (start[e] = where the event
 * starts, nmsg = the number of events) — what tfgpu_debezium_parse takes as `msgs` with these frames.
 * tfgpu_dbz_receive_registry: DoBatch with a registry.  events (HOST, events_cap entries) receives tfgpu_sr_frames' list,
 * *nevents its length (larger than events_cap: TFGPU_ERR_INVALID, nothing done).  Schema ids nobody registered are listed in
 * missing_ids (first missing_cap of *nmissing): register them and call again — nothing was parsed (*ngroups = 0).  Otherwise
 * every group of events with one schema id is parsed; event_codes[e] (HOST, events_cap entries) = TFGPU_ROW_OK, why event e
 * becomes an `_unparsed` item, TFGPU_ROW_HOST_FALLBACK (the stock code takes the whole Kafka message: every event of it carries
 * this code), or TFGPU_ROW_DROPPED: an earlier event of the same Kafka message failed and DoBuf stopped there (DoOne returns a
 * nil rest).  The groups' rows (tfgpu_dbz_receive_group; src_row and rows[i].msg = the EVENT's ordinal; events[e].msg / .index =
 * its Kafka message and QueueMessageMeta.Index) hold the events with TFGPU_ROW_OK only.                                       */
#define TFGPU_DBZ_REGISTRY_HASH 0x5343484D52454749ull
int tfgpu_debezium_compile_registry_schema(const void *schema_text, uint64_t len, tfgpu_dbz_schema **out);
int tfgpu_dbz_receiver_add_registry_schema(tfgpu_dbz_receiver *r, uint32_t schema_id, const void *schema_text, uint64_t len);
int tfgpu_debezium_registry_frames(const void *bytes, uint64_t len, int mem, const tfgpu_messages *event_msgs, const tfgpu_sr_frame *events, tfgpu_dbz_frame *frames);
int tfgpu_dbz_receive_registry(tfgpu_dbz_receiver *r, const void *bytes, uint64_t len, int mem, const tfgpu_messages *msgs,
                               tfgpu_sr_frame *events, int64_t events_cap, int64_t *nevents, int32_t *event_codes,
                               uint32_t *missing_ids, int32_t missing_cap, int32_t *nmissing, int32_t *ngroups);

/* ---- ClickHouse Native column block -------------------------------------------------------------------- */
/* The v2 ClickHouse sink turns every ChangeItem into a []any row (pkg/providers/clickhouse/async/marshaller.go:62-190) and
 * appends it to a clickhouse-go batch, which encodes ClickHouse's Native column layout.  With the batch columnar in HBM
 * the sink-side work is that layout itself:  varuint(ncols) varuint(nrows), then per column  string(name) string(type)
 * [Nullable: nrows null-map bytes] data  — fixed-width values little-endian (nil / NULL → zero), String as varuint(len) +
 * bytes.  `cols` lists the target table's columns in order, each naming a batch column and its ClickHouse type:
 * Int8…Int64, UInt8…UInt64, Float32/64, Bool, String, Date, Date32, DateTime[('tz')], DateTime64(p[, 'tz']), each
 * optionally inside Nullable(…).  The Go value must be the one the driver appends without conversion (int32 → Int32, …;
 * time.Time → the four date types; text / []byte / any → String); YT date / datetime columns are clamped to
 * [1970-01-01, 2106-01-01] first (columntypes.Restore, types.go:15-29, 92-104).  A time outside its column's range fails
 * the call (TFGPU_ERR_INVALID, the driver's DateOverflowError fails the push); other types / conversions →
 * TFGPU_ERR_UNSUPPORTED (Decimal, LowCardinality, Array, UUID, Enum, FixedString, json.Number → Float64: host).
 * The encoder is a dependency of the reference (clickhouse-go v2.46.0 / ch-go v0.71.0), not part of it: PARITY UNPINNED —
 * the oracle restates the same published format and the tests decode the block independently.                          */
typedef struct tfgpu_ch_native_column { const char *name; const char *ch_type; } tfgpu_ch_native_column;
int tfgpu_ch_native_block(const tfgpu_dbatch *in, const tfgpu_ch_native_column *cols, int32_t ncols, tfgpu_dbuf **out);

/* ---- queue serializers: pkg/serializer/queue (the "→ Kafka sink" half of configs[4]) ---- */
/* ChangeItem fields that are not columns (change_item.go:27-80).  Every array is indexed by the INPUT row of the
 * pipeline: row r of the batch reads entry src_row[r] (identity when the batch has no src_row) — the fan-in key the
 * batch carries through every row-moving step.  NULL array = the Go zero value in every row.                        */
typedef struct tfgpu_row_meta {
  const uint32_t *id;            /* ChangeItem.ID          "id"                                     */
  const uint64_t *lsn;           /* ChangeItem.LSN         "nextlsn"                                */
  const uint64_t *commit_time;   /* ChangeItem.CommitTime  "commitTime"                             */
  const int64_t *counter;        /* ChangeItem.Counter     "txPosition"                             */
  const uint32_t *tx_id_offsets; /* ChangeItem.TxID: [n+1] offsets into tx_id_data                  */
  const uint8_t *tx_id_data;
  const uint32_t *query_offsets; /* ChangeItem.Query                                                */
  const uint8_t *query_data;
  const uint8_t *names_form;     /* 0: ColumnNames / ColumnValues = the batch columns;
                                    1: ColumnNames == nil   ("columnnames":null, no "columnvalues");
                                    2: ColumnNames == []    ("columnnames":[],   no "columnvalues")  */
  int64_t n;                     /* entries in each array (rows of the pipeline input)              */
  int32_t mem;                   /* TFGPU_MEM_HOST / TFGPU_MEM_DEVICE                               */
} tfgpu_row_meta;

enum {
  TFGPU_QFMT_NATIVE = 1, /* NativeSerializer: "[" + ChangeItem.ToJSONString() + "]", batches joined by ","
                            (native_serializer.go:14-26, native_batcher.go:10-63; MarshalJSON change_item.go:568-616) */
  TFGPU_QFMT_JSON = 2    /* queue.JSONSerializer: pkg/serializer/json.go per item (UnsupportedItemKinds = update,
                            delete; AddClosingNewLine / AnyAsString off), batches joined by "\n"
                            (json_serializer.go:21-59, json_batcher.go:11-66)                                         */
};
typedef struct tfgpu_queue_options {
  int32_t format;              /* TFGPU_QFMT_*                                                       */
  int32_t batching_enabled;    /* model.Batching.Enabled                                             */
  int32_t max_change_items;    /* model.Batching.MaxChangeItems, 0 = no limit                        */
  int64_t max_message_size;    /* model.Batching.MaxMessageSize, 0 = no limit                        */
  /* NATIVE only: */
  const char *table_schema_json; /* json.Marshal(TableSchema.Columns()) rendered once by the caller, or NULL =
                                    rendered from `table_schema` / the batch's schema (TableSchema, TableName,
                                    Expression = "", FakeKey = false, no Properties)                 */
  const tfgpu_schema *table_schema;
  int32_t omit_table_schema;   /* ChangeItem.TableSchema == nil or empty                              */
  const char *const *old_key_types; /* OldKeys.KeyTypes, one per old-key column, or NULL = omitted    */
  /* rows are serialised group by group (splitByTablePartID, split.go:5-12): `ngroups` contiguous row runs of
   * `group_rows[g]` rows each; a batch never spans two groups.  NULL = one group.                    */
  int32_t ngroups;
  const int64_t *group_rows;
  /* ChangeItem.PartID is constant inside a group: one string per group, or NULL = itoa(batch part_id[r]) when the
   * batch carries part ids (sharder_transformer), "" otherwise.                                       */
  const char *const *group_part_ids;
} tfgpu_queue_options;

/* Serializer.Serialize for one table's rows: message VALUES back to back in *values; message m is the bytes
 * [msg_start[m], msg_start[m+1]) and holds rows [msg_row[m], msg_row[m+1]).  Keys are constant per call and stay with
 * the caller (Fqtn() when batching is off, nil when on).  msg_start / msg_row have room for `cap`+1 entries; the call
 * fails with TFGPU_ERR_INVALID if there are more messages (cap = nrows is always enough).
 * Rows of a non-row kind are TFGPU_ERR_UNSUPPORTED (they travel through the stock serializer); JSON format fails with
 * TFGPU_ERR_UNSUPPORTED on update / delete rows exactly where the reference returns its "unsupported kind" error.   */
/* The queue serializers whose messages ARE column bytes, and the Kafka writer's partitioner — the host halves of SURVEY §8 f4 behind
 * the ABI (round 4; they were Python):
 *   tfgpu_queue_raw_column   RawColumnSerializer.Serialize for one table / PartID (pkg/serializer/queue/raw_column_serializer.go:21-73):
 *                            a message per row = the value of `column`; the rows the reference skips with a warning (the column absent
 *                            from ColumnNames or from the TableSchema, a DataType that is neither "utf8" nor "string", a value that is
 *                            no Go string / []byte — nil included) give no message.  `schema` NULL = the batch's own.
 *   tfgpu_queue_mirror       MirrorSerializer.Serialize (mirror_serializer.go:15-52; changeitem/mirror.go:23-87): (key, value) per row =
 *                            (`sequence_key`, `data`); TFGPU_ERR_INVALID with the reference's message where it fails (not mirror items, a
 *                            TableSchema that is not RawDataSchema, a `sequence_key` that is not []byte, a nil `data`).  key_nil[r] = 1: no key.
 *   tfgpu_queue_part_groups  splitByTablePartID for one table (split.go:5-12): rows grouped by PartID, groups by first appearance.
 *   tfgpu_kafka_hash_partition / tfgpu_kafka_partitions
 *                            kafka-go's Hash balancer (vendor_patched/github.com/segmentio/kafka-go/balancer.go:153-181): FNV-1a(32) of the
 *                            key as an int32, Go's remainder, a negative result negated; -1 for a nil key (round robin there).  The second
 *                            form hashes a whole key column on the device.
 * *values / *keys: the bytes back to back in HBM (the column's own buffer when every row has a value); msg_start / key_start: HOST arrays of
 * cap + 1 offsets, message m = [msg_start[m], msg_start[m + 1]).                                                                          */
int tfgpu_queue_raw_column(const tfgpu_dbatch *b, const char *column, const tfgpu_schema *schema, tfgpu_dbuf **values, uint32_t *msg_start, int64_t cap, int64_t *nmsg);
int tfgpu_queue_mirror(const tfgpu_dbatch *b, const tfgpu_schema *schema, tfgpu_dbuf **values, uint32_t *msg_start, tfgpu_dbuf **keys, uint32_t *key_start,
                       uint8_t *key_nil, int64_t cap, int64_t *nmsg);
int tfgpu_queue_part_groups(const tfgpu_dbatch *b, int32_t *order, int64_t *group_rows, uint32_t *group_part_id, int64_t cap, int64_t *ngroups);
int32_t tfgpu_kafka_hash_partition(const void *key, int64_t len, int32_t npartitions);
int tfgpu_kafka_partitions(const tfgpu_dbatch *b, const char *key_column, int32_t npartitions, int32_t *partitions);
int tfgpu_queue_serialize(const tfgpu_queue_options *opts, const tfgpu_dbatch *b, const tfgpu_row_meta *meta,
                          tfgpu_dbuf **values, uint64_t *msg_start, int64_t *msg_row, int64_t cap, int64_t *nmsg);

/* ---- Debezium emitter: queue.DebeziumSerializer (the "debezium" sink format of configs[4]) ----
 * Emitter.EmitKV for every row of one table's batch (pkg/serializer/queue/debezium_serializer.go:26-43,
 * pkg/debezium/emitter_value_converter.go:574-690): 0..3 messages per ChangeItem — one for an insert or a plain update, (delete,
 * tombstone) for a delete, (delete, tombstone, insert) for an update whose primary key changed (ChangeItem.KeysChanged), the
 * tombstones dropped under tombstones.on.delete=false.  Message m: key = keys[key_start[m], key_start[m+1]) (empty with
 * drop_keys), value = values[val_start[m], val_start[m+1]) or nil when val_null[m] (a tombstone); msg_row[m] = the batch row it
 * came from.  Every array has room for cap+1 (starts) / cap entries; cap = 3 * nrows is always enough.
 * Both halves are PackerIncludeSchema's {"payload":…,"schema":…} (packer/packer_include_schema.go:14-40) — or the payload alone under
 * key / value.converter.schemas.enable=false (PackerSkipSchema) — Go maps marshalled by util.JSONMarshalUnescape, members in byte
 * order, no HTML escaping; a schema-registry URL / YSR namespace selects packers that stay with the host (packer/factory.go:13-98).  `table_schema` is TableSchema.Columns() with OriginalType
 * and the PrimaryKey flags (the batch itself carries neither); columns of the schema missing from the batch are TOASTed
 * (unavailable.value.placeholder).  Row meta: id → source.txId, lsn → source.lsn, commit_time → both ts_ms (NULL = zeros).
 * An `any` value (TFGPU_R_JSON) is taken as what the ABI says it is — json.Marshal's text of the Go value: compact, object members in key order, HTML bytes not
 * escaped — so marshalling it again (pg:json, hstore maps, arrays) is the identity and the emitter copies it; a producer that keeps source text there must canonicalise it.
 * Parameters: the format settings map (parameters.go:140-215 fills the defaults): database.dbname, topic.prefix, dt.source.type
 * ("" | "pg" | "ydb" | "mysql"), decimal.handling.mode (precise | string), tombstones.on.delete, dt.add.original.type.info,
 * unavailable.value.placeholder, dt.unknown.types.policy.
 * Device-resident Postgres types (pkg/debezium/pg/emitter.go:262-629): boolean, bit(1), smallint, integer, bigint, oid, real, double
 * precision, text / character* / uuid / cidr / macaddr / citext / int4range / int8range / daterange, inet, bytea, date and
 * timestamp[(p)] with / without time zone as time.Time, time[(p)] with / without time zone, json / jsonb, hstore as a map, xml,
 * numeric[(p,s)] up to 38 digits, money, bit(n) / bit varying(n), point, interval, tsrange, numrange and tstzrange in their plain
 * two-bound forms; and the ydb: types (pkg/debezium/ydb/emitter.go:123-232) with dt.source.type = "ydb" (source.txId = the row meta's
 * TxID, source.step = CommitTime) and the mysql: types (pkg/debezium/mysql/emitter.go:168-388) with dt.source.type = "mysql" (source.db =
 * the schema, file / pos from the LSN, gtid = TxID).  Postgres enums (Properties[pg:enum_all_values]) and arrays
 * (every element type but `timestamp without time zone`).  Anything else — arrays of arrays, binaries / bits given as base64 text, hstore / range / time texts only pgtype's parsers decide,
 * the schema-registry packers, a value of a Go type the device does not convert — is refused BY NAME
 * with TFGPU_ERR_UNSUPPORTED and travels through the stock emitter; where the reference itself returns an error (an unknown
 * type under policy "fail", a json.Number that is no integer, "unknown type of value") the call fails with TFGPU_ERR_INVALID.  */
typedef struct tfgpu_dbz_emit_options {
  const char *const *param_keys;    /* connector parameters: nparams (key, value) pairs                  */
  const char *const *param_values;
  int32_t nparams;
  const char *version;              /* source.version; NULL = "1.1.2.Final" (debezium_serializer.go:118) */
  int32_t drop_keys;                /* dropKeys: no message keys                                         */
  int32_t snapshot;                 /* isSnapshot: inserts are op "r", source.snapshot = "true"          */
  const tfgpu_schema *table_schema; /* required                                                          */
} tfgpu_dbz_emit_options;
int tfgpu_debezium_emit(const tfgpu_dbz_emit_options *opts, const tfgpu_dbatch *b, const tfgpu_row_meta *meta, tfgpu_dbuf **keys, uint64_t *key_start,
                        tfgpu_dbuf **values, uint64_t *val_start, uint8_t *val_null, int64_t *msg_row, int64_t cap, int64_t *nmsg);

/* ---- profiling hooks (bench.py / rocprof cross-check) ------------------- */
/* Per-kernel accumulated device time measured with HIP events on the library
 * stream.  Enable, run, then read back name/launches/total_ms.              */
int tfgpu_prof_enable(int on);
int tfgpu_prof_reset(void);
int tfgpu_prof_count(void);
int tfgpu_prof_get(int i, const char **name, int64_t *launches, double *total_ms);
/* The units (rows for the row kernels) entry i's timed launches were issued over, summed; 0 when the call site states
 * none.  bench.py prices a kernel against the rows it was LAUNCHED on (a mask behind a hoisted filter sees the kept rows,
 * tf_transform.hip chain_sequence), not the rows of the batch that entered the chain.                                  */
int tfgpu_prof_get_units(int i, int64_t *units);

#ifdef __cplusplus
}
#endif
#endif /* TFGPU_H */
