/* cli_keys.h - pause / resume keys, Ctrl-C.
   Part of the one translation unit ecloop_hip_cli.c (included there, in this order). */
/* ------------------------------------------------------------------------------------------- pause / resume keys */
/* 'p' parks the device threads at their next progress report, 'r' lets them go on (main.c:874-888; the reference's raw
   /dev/tty listener is utils.c:546-624).  Keys come from the controlling terminal in non-canonical mode, or from the
   path in ECLOOP_HIP_TTY (a FIFO works: containers without ptys); without either nothing is installed.  One detached
   thread polls the descriptor; the terminal's settings are put back at exit. */
static struct { int fd; bool is_terminal; struct termios saved; report_t *rep; } keys = {-1, false, {0}, NULL};
static void keys_restore(void) {
  if (keys.fd < 0) return;
  if (keys.is_terminal) tcsetattr(keys.fd, TCSANOW, &keys.saved);
  close(keys.fd), keys.fd = -1;
}
static void *keys_thread(void *unused) {
  (void)unused;
  struct pollfd p = {keys.fd, POLLIN, 0};
  for (char key; p.fd >= 0 && poll(&p, 1, 200) >= 0; p.fd = keys.fd)
    if ((p.revents & POLLIN) && read(p.fd, &key, 1) == 1 && (key == 'p' || key == 'r')) report_pause(keys.rep, key == 'p');
  return NULL;
}
static void keys_listen(report_t *rep) {
  const char *path = getenv("ECLOOP_HIP_TTY");
  keys.fd = open(path ? path : "/dev/tty", (path ? O_RDWR : O_RDONLY) | O_NONBLOCK);
  if (keys.fd < 0) return;
  keys.rep = rep;
  keys.is_terminal = tcgetattr(keys.fd, &keys.saved) == 0;
  if (!keys.is_terminal && !path) { close(keys.fd), keys.fd = -1; return; }
  atexit(keys_restore);
  if (keys.is_terminal) {
    struct termios t = keys.saved;
    t.c_lflag &= ~(tcflag_t)(ICANON | ECHO);
    tcsetattr(keys.fd, TCSANOW, &t);
  }
  pthread_t th;
  if (!pthread_create(&th, NULL, keys_thread, NULL)) pthread_detach(th);
}
static void on_sigint(int sig) { /* main.c:867-872: what was printed so far reaches its destination, then out */
  fflush(stderr), fflush(stdout);
  fputc('\n', stdout);
  exit(sig);
}
