"""Matchers behind the glue-factory plugin surface (dict in / dict out)."""
