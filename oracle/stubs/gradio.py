"""ORACLE TEST INFRASTRUCTURE — stand-in for `gradio` (utils/parse.py:8 only uses gr.Error)."""


class Error(Exception):
    pass
