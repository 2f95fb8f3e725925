"""Host mirror of the transformation sink middleware (pkg/transformer/transformation.go:122-235,
pkg/middlewares/transformation.go:12-34) above the C ABI — what the Go shim's `Push` does around the device:

    items ──SplitByTableID──► per table, in order ──schema-hash cut──► runs ──tfgpu_transformation_push──► transformed
                                                                                   └─► TransformerErrors ──► `__transform_error` rows

A *run* here is a DeviceBatch (one table, one schema) or a control item (a non-row ChangeItem such as InitTableLoad: it has
no columns, no transformer changes it, it is passed through in place).  The stage keeps the reference's contracts: runs of
one table stay in order, tables may interleave in any order (Go ranges over a map), errors go to the sink with one more
utf8 column `__transform_error` — or are logged and dropped when ErrorsOutput is devnull (transformation.go:173-190)."""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import abi, lib

TRANSFORM_ERROR_COLUMN = "__transform_error"  # transformation.go:18

# Go's error text is the shim's to write (it holds the error values); the mirror names the reason by its code
ERROR_TEXT = {  # filter_rows.go:99-125, 145-176, util.go:65-68
    "UNSUPPORTED_KIND": "fatal: FilterRowsTransformer can be applied to insert items only",
    "COLUMN_NOT_FOUND": "fatal: unable to find the filter's column in the item",
    "INT_OVERFLOW": "fatal: uint64 value does not fit int64",
    "TYPE_PAIR": "fatal: the value's type cannot be compared with the filter's",
}


class ControlItem:
    """A non-row ChangeItem (InitShardedTableLoad, InitTableLoad, DoneTableLoad, ...): TableID and Kind only."""

    def __init__(self, ns: str, table: str, kind: str):
        self.ns, self.table, self.kind = ns, table, kind

    def table_id(self):
        return (self.ns, self.table)


Run = Union[lib.DeviceBatch, ControlItem]


def _table_id(run: Run) -> Tuple[str, str]:
    if isinstance(run, ControlItem):
        return run.table_id()
    v = run.view()
    return ((v.table_ns or b"").decode(), (v.table_name or b"").decode())


def split_by_table_id(runs: Sequence[Run]):
    """abstract.SplitByTableID (pkg/abstract/changeitem/utils.go:130-136): table → its runs, order kept per table."""
    out = {}
    for r in runs:
        out.setdefault(_table_id(r), []).append(r)
    return out


def error_change_items(err_batch: lib.DeviceBatch, reasons: Sequence[str]) -> abi.Batch:
    """errorChangeItems (transformation.go:191-235): the refused items with ColumnNames + `__transform_error`,
    ColumnValues + the error text, and the TableSchema extended by that utf8 column."""
    b = err_batch.download()
    assert b.nrows == len(reasons)
    text = [r.encode("utf-8") for r in reasons]
    offs = np.zeros(len(text) + 1, np.uint32)
    if text:
        offs[1:] = np.cumsum([len(t) for t in text])
    col = abi.Column(TRANSFORM_ERROR_COLUMN, "utf8", abi.R_STRING, offsets=offs, data=np.frombuffer(b"".join(text), np.uint8).copy())
    b.cols.append(col)
    sch = err_batch.table_schema()
    base = list(sch.cols) if sch is not None else [abi.ColSchema(c.name, c.dtype, False, "", "") for c in b.cols[:-1]]
    b.schema = abi.Schema(base + [abi.ColSchema(TRANSFORM_ERROR_COLUMN, "utf8", False, "", "")])
    return b


class Stage:
    """`transformation` as a sink middleware: push(runs) → sink(list of transformed runs / error batches)."""

    def __init__(self, transformers: Sequence[lib.Transformer], sink: Callable[[List], None], errors_output: Optional[str] = None):
        if errors_output not in (None, "sink", "devnull"):
            raise ValueError("output format %s not implemented" % errors_output)  # transformation.go:187
        self.t = lib.Transformation(transformers)
        self.sink, self.errors_output = sink, errors_output
        self.dropped_errors = 0

    def push(self, runs: Sequence[Run]):
        transformed: List = []
        error_items: List[abi.Batch] = []
        for _tid, table_runs in split_by_table_id(runs).items():  # one goroutine per table in the reference (:131-135)
            for run in table_runs:                                 # transformation.do: runs of one schema, in order
                if isinstance(run, ControlItem):
                    transformed.append(run)
                    continue
                res = self.t.push_run(run)
                transformed.append(res.transformed)
                k = 0
                for step, eb in res.error_batches:
                    n = eb.nrows
                    reasons = [ERROR_TEXT.get(e[1], e[1]) for e in res.errors[k:k + n]]
                    assert all(e[2] == step for e in res.errors[k:k + n])
                    k += n
                    error_items.append(error_change_items(eb, reasons))
        if error_items:  # pushErrors :173-190
            if self.errors_output == "devnull":
                self.dropped_errors += sum(b.nrows for b in error_items)
            else:
                self.sink(error_items)
        self.sink(transformed)

    def stats(self):
        return self.t.stats()
