"""String and time helpers that define on-disk names and console output.

``normalise_string`` determines model directory names (``model.name``), so it
reproduces the behaviour of ``scvae/utilities.py:63-76`` exactly; the duration
format follows ``scvae/utilities.py:36-60`` because the training loop prints it.
"""

import re
import time
from math import floor

_TO_UNDERSCORE = " -/"
_TO_NOTHING = "(),$<>:\"/\\|?*"


def normalise_string(s):
    s = s.lower()
    s = re.sub("[" + re.escape(_TO_UNDERSCORE) + "]", "_", s)
    s = re.sub("[" + re.escape(_TO_NOTHING) + "]", "", s)
    return s


def capitalise_string(s):
    head, space, tail = s.partition(" ") if " " in s else (s, "", "")
    if not re.match(r"[A-Z]", head):
        head = head.capitalize()
    return head + space + tail


def enumerate_strings(strings, conjunction="and"):
    if not isinstance(strings, list):
        raise ValueError("`strings` should be a list of strings.")
    conjunction = conjunction.strip()
    if len(strings) == 1:
        return strings[0]
    if len(strings) == 2:
        return " {} ".format(conjunction).join(strings)
    if len(strings) >= 3:
        return "{}, {} {}".format(", ".join(strings[:-1]), conjunction,
                                  strings[-1])
    raise ValueError("`strings` does not contain any strings.")


def format_time(t):
    return time.strftime("%Y-%m-%d %H:%M:%S %Z", time.localtime(t))


def format_duration(seconds):
    if seconds < 0.001:
        return "<1 ms"
    if seconds < 1:
        return "{:.0f} ms".format(1000 * seconds)
    if seconds < 60:
        return "{:.3g} s".format(seconds)
    hours = floor(seconds / 3600)
    minutes = floor((seconds / 60) % 60)
    rest = seconds % 60
    if round(rest) == 60:
        rest = 0
        minutes += 1
    if seconds < 3600:
        return "{:.0f}m {:.0f}s".format(minutes, rest)
    if minutes == 60:
        minutes = 0
        hours += 1
    return "{:.0f}h {:.0f}m {:.0f}s".format(hours, minutes, rest)


def heading(string, underline_symbol="-", plain=False):
    return "{}\n{}\n".format(string, len(string) * underline_symbol)


def title(string):
    return heading(string, underline_symbol="=")


def subtitle(string):
    return heading(string, underline_symbol="-")
