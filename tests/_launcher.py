"""A process that never touches the GPU and starts other programs on request (one JSON request per stdin line, one JSON reply per
stdout line).  tests/conftest.py starts it at session start -- before anything in the pytest process can have initialised HIP -- so
that a test may launch the rank processes of bench.py at ANY point of the session: on the GPU pool a process that has initialised
the GPU must not fork + exec, and the pytest process has as soon as the first GPU test ran."""
import json
import subprocess
import sys


def main():
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        req = json.loads(line)
        try:
            p = subprocess.run(req["cmd"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=req.get("timeout", 900),
                               env=req.get("env"), cwd=req.get("cwd"))
            rep = {"rc": p.returncode, "stdout": p.stdout, "stderr": p.stderr}
        except subprocess.TimeoutExpired as e:
            rep = {"rc": -9, "stdout": (e.stdout or b"").decode(errors="replace") if isinstance(e.stdout, bytes) else (e.stdout or ""),
                   "stderr": "timeout after %s s" % req.get("timeout", 900)}
        except Exception as e:                     # report, keep serving
            rep = {"rc": -1, "stdout": "", "stderr": repr(e)}
        sys.stdout.write(json.dumps(rep) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
