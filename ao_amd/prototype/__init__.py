"""Prototype-namespace mirrors (torchao/prototype/*) that sit on the SURVEY.md section 8 path."""
