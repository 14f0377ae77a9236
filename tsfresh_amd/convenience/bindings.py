"""dask / spark partition bindings (reference: tsfresh/convenience/bindings.py:9-60, :62-162, :164-264).

The reference hands dask one `(id, kind)` group at a time: `df.groupby([id, kind]).apply(_feature_extraction_on_chunk_
helper)` runs the Python dispatcher per series and returns long `(id, variable, value)` rows.  One series per call is
the wrong grain for a GPU: here the unit of work is a whole PARTITION -- `feature_extraction_on_partition` packs every
(id, kind) series of a pandas frame into one ragged batch per kind, extracts it in one pass and returns the same long
rows; `dask_feature_extraction_on_chunk` maps it over the partitions of a dask DataFrame (which must be partitioned
so that no id is split across partitions, e.g. `df.set_index(column_id)` or `shuffle(on=column_id)`).

`spark_feature_extraction_on_chunk` does the same for a Spark DataFrame (`repartition(column_id)`, then one grouped-map
pandas call per Spark partition).

Neither dask nor pyspark is a dependency of this package: the entry points import them lazily and say so when they
are missing.
"""
import numpy as np
import pandas as pd

from tsfresh_amd.feature_extraction.extraction import extract_features
from tsfresh_amd.feature_extraction.settings import ComprehensiveFCParameters


def feature_extraction_on_partition(df, column_id, column_kind, column_value, column_sort=None,
                                    default_fc_parameters=None, kind_to_fc_parameters=None, device=None):
    """All series of one pandas frame (a dask partition) -> long DataFrame `[column_id, "variable", "value"]` with the
    rows `_feature_extraction_on_chunk_helper` (bindings.py:9-60) would produce for every (id, kind) group: `variable` is
    the reference's column name `"{kind}__{calculator}__{params}"`, `value` a float64."""
    if default_fc_parameters is None and kind_to_fc_parameters is None:
        default_fc_parameters = ComprehensiveFCParameters()
    elif default_fc_parameters is None and kind_to_fc_parameters is not None:
        default_fc_parameters = {}
    empty = pd.DataFrame({column_id: pd.Series([], dtype=df[column_id].dtype if column_id in df else "int64"),
                          "variable": pd.Series([], dtype=object), "value": pd.Series([], dtype="float64")})
    if len(df) == 0:
        return empty
    tuples = extract_features(df, column_id=column_id, column_kind=column_kind, column_value=column_value,
                              column_sort=column_sort, default_fc_parameters=default_fc_parameters,
                              kind_to_fc_parameters=kind_to_fc_parameters, pivot=False, device=device)
    if not tuples:
        return empty
    ids, variables, values = zip(*tuples)
    out = pd.DataFrame({column_id: np.asarray(ids), "variable": np.asarray(variables, dtype=object),
                        "value": np.asarray(values, dtype=np.float64)})
    return out[[column_id, "variable", "value"]]


def dask_feature_extraction_on_chunk(df, column_id, column_kind, column_value, column_sort=None,
                                     default_fc_parameters=None, kind_to_fc_parameters=None, device=None):
    """The reference's dask entry point (bindings.py:62) at partition grain.

    :param df: a dask DataFrame in long format with the columns `column_id`, `column_kind`, `column_value` (and
        `column_sort`), partitioned so that every id lives in ONE partition.  (The reference takes the frame already
        grouped by (id, kind); a `DataFrameGroupBy` is accepted too and its underlying frame is used.)
    :return: a dask DataFrame with the columns `[column_id, "variable", "value"]`, as the reference returns; pivot it
        with `pivot_table(index=column_id, columns="variable", values="value")`."""
    try:
        import dask.dataframe as dd  # noqa: F401
    except ImportError as e:  # pragma: no cover - dask is not installed in the build image
        raise ImportError("dask_feature_extraction_on_chunk needs dask[dataframe]; for a pandas frame call "
                          "feature_extraction_on_partition directly") from e
    frame = getattr(df, "obj", df)  # a groupby object carries its frame as .obj
    meta = pd.DataFrame({column_id: pd.Series([], dtype=frame[column_id].dtype), "variable": pd.Series([], dtype=object),
                         "value": pd.Series([], dtype="float64")})
    return frame.map_partitions(feature_extraction_on_partition, column_id=column_id, column_kind=column_kind,
                                column_value=column_value, column_sort=column_sort,
                                default_fc_parameters=default_fc_parameters, kind_to_fc_parameters=kind_to_fc_parameters,
                                device=device, meta=meta)


def spark_feature_extraction_on_chunk(df, column_id, column_kind, column_value, column_sort=None,
                                      default_fc_parameters=None, kind_to_fc_parameters=None, device=None):
    """The reference's Spark entry point (bindings.py:164) at partition grain.

    The reference registers `_feature_extraction_on_chunk_helper` as a GROUPED_MAP pandas UDF and applies it to every
    (id, kind) group (bindings.py:252-264).  Here the frame is repartitioned by `column_id` (every id in ONE partition)
    and each Spark partition is extracted in one pass on the executor's GPU.

    :param df: a `pyspark.sql.DataFrame` in long format with the columns `column_id`, `column_kind`, `column_value`
        (and `column_sort`).  (The reference takes `df.groupby([id, kind])`; a `GroupedData` cannot be un-grouped, so
        pass the frame itself.)
    :return: `pyspark.sql.DataFrame[column_id, variable: string, value: double]`, as the reference returns; pivot it
        with `.groupby(column_id).pivot("variable").sum("value")`."""
    try:
        from pyspark.sql import functions as F
        from pyspark.sql.types import DoubleType, StringType, StructField, StructType
    except ImportError as e:  # pragma: no cover - pyspark is not installed in the build image
        raise ImportError("spark_feature_extraction_on_chunk needs pyspark; for a pandas frame call "
                          "feature_extraction_on_partition directly") from e
    if not hasattr(df, "repartition"):
        raise TypeError("pass the Spark DataFrame itself (not df.groupby(...)): the partitions are formed by column_id here")
    id_field = [f for f in df.schema.fields if f.name == column_id]
    if not id_field:
        raise ValueError("column {!r} is not in the Spark DataFrame".format(column_id))
    schema = StructType([StructField(column_id, id_field[0].dataType), StructField("variable", StringType()),
                         StructField("value", DoubleType())])

    def on_partition(pdf):
        return feature_extraction_on_partition(pdf.drop(columns=["__tsfa_part"]), column_id=column_id,
                                               column_kind=column_kind, column_value=column_value, column_sort=column_sort,
                                               default_fc_parameters=default_fc_parameters,
                                               kind_to_fc_parameters=kind_to_fc_parameters, device=device)

    parts = df.repartition(column_id).withColumn("__tsfa_part", F.spark_partition_id())
    return parts.groupby("__tsfa_part").applyInPandas(on_partition, schema=schema)
