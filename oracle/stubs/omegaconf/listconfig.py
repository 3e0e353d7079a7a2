"""ListConfig stand-in (test infrastructure only; see package docstring)."""


class ListConfig(list):
    pass
