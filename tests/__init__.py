"""Test suite of the MI355X step engine (a regular package: /root/reference has a `tests` package of its own on sys.path when the reference is imported)."""
