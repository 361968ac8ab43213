"""Empty stand-in: the reference imports openpyxl at module import
(utilities/Utility_Class.py:5) but the environment path never calls it."""


class Workbook(object):
    def __init__(self, *a, **k):
        raise RuntimeError("openpyxl stand-in: not available in this image")


def load_workbook(*a, **k):
    raise RuntimeError("openpyxl stand-in: not available in this image")
