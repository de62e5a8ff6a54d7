"""ORACLE TEST INFRASTRUCTURE — stand-in for `inflect` (utils/parse.py:7,11 builds an engine at import)."""


class engine:
    def plural_noun(self, w, count=None):
        return w + "s"

    plural = plural_noun

    def singular_noun(self, w):
        return w[:-1] if w.endswith("s") else False

    def number_to_words(self, n):
        return str(n)

    def a(self, w):
        return ("an " if w[:1].lower() in "aeiou" else "a ") + w
