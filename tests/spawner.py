"""A helper process that starts other processes on request.  tests/conftest.py launches it when pytest starts -- BEFORE
anything has initialised the GPU -- so that multi-process GPU tests get their ranks as children of a process that never
touched the device (on the GPU box a process that has initialised HIP must not fork + exec).
Protocol: one JSON request per line on stdin {"cmds": [[argv...], ...], "env": {...}, "timeout": s}; one JSON reply per
line on stdout {"rc": [...], "out": [...]} after all commands of the request (started together) have ended."""
import json
import os
import subprocess
import sys


def main():
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        req = json.loads(line)
        env = dict(os.environ)
        env.update(req.get("env", {}))
        procs = [subprocess.Popen(c, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for c in req["cmds"]]
        rc, out = [], []
        for p in procs:
            try:
                o = p.communicate(timeout=req.get("timeout", 300))[0]
            except subprocess.TimeoutExpired:
                p.kill()
                o = p.communicate()[0] + "\n[spawner] timeout"
            rc.append(p.returncode)
            out.append(o[-6000:])
        sys.stdout.write(json.dumps({"rc": rc, "out": out}) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
