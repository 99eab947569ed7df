"""The two path helpers of the reference's libutil.py that the synthesis path uses
(safe_makedir 12-14, basename 46-49)."""
import os


def safe_makedir(d):
    if not os.path.isdir(d):
        os.makedirs(d)


def basename(fname):
    """File name without directory and without its last extension."""
    name = os.path.split(fname)[1]
    stem, dot, ext = name.rpartition(".")
    return stem if (dot and ext != "") else name
