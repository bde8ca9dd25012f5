"""Identity stamps for the summaries under profiles/: the build the profiled run loaded (the `build` object of the bench line
inside the profiler's log: sha256 of the shared library and of its sources, computed on the GPU box by the profiled process)
and the commit the summary is filed under (git HEAD here, with a dirty marker when tracked files differ from it)."""
import json
import re
import subprocess


def from_log(path):
    """The `build` object of the bench.py JSON line found in a profiler log, or {}."""
    try:
        for line in open(path, errors="replace"):
            if line.startswith("{") and '"build"' in line:
                return json.loads(line).get("build", {})
    except (OSError, ValueError):
        pass
    return {}


def git_head():
    try:
        head = subprocess.run(["git", "rev-parse", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
        # (the summaries themselves are being rewritten while this runs: profiles/ does not count)
        dirty = subprocess.run(["git", "status", "--porcelain", "--untracked-files=no", "--", ".", ":!profiles"],
                               capture_output=True, text=True).stdout.strip()
        return head + ("+dirty" if dirty else "")
    except Exception:
        return None


def stamp(out: dict, log_path: str) -> dict:
    b = from_log(log_path)
    out["source_sha256"] = b.get("source_sha256")
    out["lib_sha256"] = b.get("lib_sha256")
    out["git_head_at_summary"] = git_head()
    return out
