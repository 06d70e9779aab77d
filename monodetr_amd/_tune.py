"""``MDETR_TUNE="key=value,key=value"``: the one environment variable through which tests force a launch geometry or an alternative
route (csrc/mdetr_tune.h reads the same variable in the launchers).  Nothing in the product sets it."""
import os


def get(key, default=None):
    for item in os.environ.get("MDETR_TUNE", "").split(","):
        k, sep, v = item.partition("=")
        if sep and k.strip() == key:
            return v.strip()
    return default
