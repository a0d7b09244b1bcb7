"""Caption normalisation of the TTA datasets (reference src/data/utils.py:32-72)."""
import re

_PUNCT = re.compile(r"([.!\"()*#:;~])")
_SPACES = re.compile(r"\s{2,}")


def pre_caption(caption, max_words=50):
    """Lower-case, replace . ! " ( ) * # : ; ~ by a blank, collapse runs of white space, strip trailing newlines and
    outer blanks, keep the first ``max_words`` blank-separated words.  A list of strings is joined with blanks first.
    Anything else is returned unchanged (as the reference does)."""
    if isinstance(caption, list):
        caption = " ".join(caption)
    elif not isinstance(caption, str):
        return caption
    caption = _PUNCT.sub(" ", caption.lower())
    caption = _SPACES.sub(" ", caption)
    caption = caption.rstrip("\n").strip(" ")
    words = caption.split(" ")
    if len(words) > max_words:
        caption = " ".join(words[:max_words])
    return caption
