import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from transferia_amd import abi, confluent_sr, lib
from oracle import oracle
lib.init()
schema = '{"title":"a.b","type":"object","properties":{"n":{"type":"number"},"s":{"type":"string"},"x":{"type":"array"},"i":{"type":"integer"}}}'
o = confluent_sr.sr_json_options(3, schema)
pay = [b'{"n":1.5,"s":"abc","x":["m","t","Fe",""],"i":7}', b'{"n":2,"s":"q","x":[1,2],"i":8}', b'{"x":{"a":1},"n":3}', b'{"s":"zz"}']
for k in (1, 2, 4):
    data, cm = abi.messages([b"\0\0\0\0\x03" + p for p in pay[:k]])
    got = lib.sr_json_parse(o, data, cm)
    ref = oracle.sr_json_parse(o, data, cm)
    print("k", k, "errors", got.errors, ref.errors)
    for c, rc in zip(got.batch.cols, ref.batch.cols):
        print(" ", c.name, c.dtype, c.repr, "valid", None if c.validity is None else c.validity.astype(int).tolist(),
              "off", None if c.offsets is None else c.offsets.tolist(), "data", None if c.data is None else bytes(c.data), "vals", None if c.values is None else c.values.tolist())
        print("   ref", rc.repr, None if rc.offsets is None else rc.offsets.tolist(), None if rc.data is None else bytes(rc.data), None if rc.values is None else rc.values.tolist())
    print("  src_row", got.batch.src_row, "part", got.batch.part_id)
