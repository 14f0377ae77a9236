"""JSON codec of pandas containers / result frames for the frame-level fixtures (gen_golden_frames.py writes, the tests
read).  Values round-trip exactly (Python's json writes float repr); ids keep their type: int, str or tuple."""
import numpy as np
import pandas as pd


def _enc_scalar(v):
    if isinstance(v, tuple):
        return {"tuple": [_enc_scalar(u) for u in v]}
    if isinstance(v, (np.integer,)):
        return int(v)
    if isinstance(v, (np.floating,)):
        return float(v)
    if isinstance(v, (np.bool_,)):
        return bool(v)
    if isinstance(v, pd.Timestamp):
        return {"timestamp": v.isoformat()}
    return v


def _dec_scalar(v):
    if isinstance(v, dict) and "tuple" in v:
        return tuple(_dec_scalar(u) for u in v["tuple"])
    if isinstance(v, dict) and "timestamp" in v:
        return pd.Timestamp(v["timestamp"])
    return v


def encode_frame(df):
    return {"columns": [str(c) for c in df.columns], "column_dtypes": [str(t) for t in df.dtypes],
            "index": [_enc_scalar(v) for v in df.index.tolist()], "index_dtype": str(df.index.dtype),
            "index_name": df.index.name,
            "values": [[_enc_scalar(v) for v in col] for col in (df[c].tolist() for c in df.columns)]}


def decode_frame(d):
    data = {}
    for c, t, col in zip(d["columns"], d["column_dtypes"], d["values"]):
        vals = [_dec_scalar(v) for v in col]
        if t == "object":
            s = pd.Series(vals, dtype=object)
        else:
            s = pd.Series(vals).astype(t)
        data[c] = s
    df = pd.DataFrame(data, columns=d["columns"])
    idx = [_dec_scalar(v) for v in d["index"]]
    if d["index_dtype"] == "object":
        df.index = pd.Index(idx, dtype=object, name=d["index_name"])
    else:
        df.index = pd.Index(idx, name=d["index_name"]).astype(d["index_dtype"])
    return df


def encode_container(c):
    if isinstance(c, dict):
        return {"dict": {str(k): encode_frame(v) for k, v in c.items()}}
    return {"frame": encode_frame(c)}


def decode_container(d):
    if "dict" in d:
        return {k: decode_frame(v) for k, v in d["dict"].items()}
    return decode_frame(d["frame"])
