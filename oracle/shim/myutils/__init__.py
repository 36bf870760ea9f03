"""regular package so it shadows the empty namespace dir /root/reference/src/myutils."""
