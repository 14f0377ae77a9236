"""Column-name <-> parameter-dict conversion.

Same contract as the reference's tsfresh/utilities/string_manipulation.py (convert_to_output_format :47,
get_config_from_string :10): the names produced here must match the reference byte for byte, because users select
features by column name (settings.from_columns) and the parity tests compare by name.
"""
import ast

import numpy as np


def convert_to_output_format(param):
    """``{"b": 1, "a": "x"}`` -> ``'a_"x"__b_1'``: keys sorted, string values wrapped in double quotes."""
    parts = []
    for key in sorted(param.keys()):
        value = param[key]
        text = '"' + str(value) + '"' if isinstance(value, str) else str(value)
        parts.append(str(key) + "_" + text)
    return "__".join(parts)


def get_config_from_string(parts):
    """Inverse of :func:`convert_to_output_format` for a column name already split on ``"__"``.

    ``parts[0]`` is the kind, ``parts[1]`` the calculator; the rest are ``<key>_<value>`` items whose value is
    parsed as a Python literal (``nan``/``inf``/``-inf`` handled explicitly).  Returns None without parameters.
    """
    items = parts[2:]
    if not items:
        return None
    config = {}
    for item in items:
        key, value = item.rsplit("_", 1)
        lowered = value.lower()
        if lowered == "nan":
            config[key] = np.nan
        elif lowered == "-inf":
            config[key] = -np.inf
        elif lowered == "inf":
            config[key] = np.inf
        else:
            config[key] = ast.literal_eval(value)
    return config
