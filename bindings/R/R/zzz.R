# bindings/R/R/zzz.R -- load libgpvecchia_hip.so beside the package's own shared object.
# NOT RUN in the repository that ships it (no R in its images); written against R's documented API.
# Reference: NAMESPACE:29 (useDynLib(GPvecchia)) stays; this adds the HIP library for the .C() stubs of
# RcppExports_hip.R and registers the .Call shim of src/gpvR_plan.c.

.gpv_env <- new.env(parent = emptyenv())

.onLoad <- function(libname, pkgname) {
  path <- Sys.getenv("GPVECCHIA_HIP_LIB", unset = "libgpvecchia_hip.so")
  .gpv_env$dll <- tryCatch(dyn.load(path, local = FALSE, now = TRUE), error = function(e) NULL)
  # the .Call shim for the plan handle (src/gpvR_plan.c, its own small shared object beside the library)
  shim <- Sys.getenv("GPVECCHIA_HIP_SHIM", unset = file.path(dirname(path), paste0("gpvR_plan", .Platform$dynlib.ext)))
  .gpv_env$shim <- if (is.null(.gpv_env$dll)) NULL else tryCatch(dyn.load(shim), error = function(e) NULL)
  .gpv_env$have_hip <- !is.null(.gpv_env$dll) && gpv_device_count() > 0L
  if (is.null(.gpv_env$dll))
    packageStartupMessage("GPvecchia: ", path, " not found; U_NZentries stays on the package's own OpenMP path")
}

.onUnload <- function(libpath) {
  if (!is.null(.gpv_env$dll)) {
    try(.C("gpv_plan_cache_clear"), silent = TRUE)      # the literal drop-in keeps ONE device plan between calls
    if (!is.null(.gpv_env$shim)) try(dyn.unload(.gpv_env$shim[["path"]]), silent = TRUE)
    try(dyn.unload(.gpv_env$dll[["path"]]), silent = TRUE)
  }
}

# number of HIP devices the library sees (0: no GPU; the library has no CPU fallback and says so by status code)
# (both through the .Call shim of src/gpvR_plan.c: gpv_device_count / gpv_last_hip_error return their status as the C
# function value, which .C() cannot see)
gpv_device_count <- function() {
  if (is.null(.gpv_env$dll)) return(0L)
  if (is.null(.gpv_env$shim)) return(1L)        # no shim: assume a device; the literal drop-ins report status 1 if there is none
  .Call("gpvR_device_count", PACKAGE = "gpvR_plan")
}

# text behind a status 6 (GPV_ERR_HIP): "hipErrorOutOfMemory: out of memory [hipMalloc(...), gpv_api.hip:NNN]"
gpv_last_error <- function() if (is.null(.gpv_env$shim)) "" else .Call("gpvR_last_error", PACKAGE = "gpvR_plan")

.gpv_check <- function(status, what) {
  if (status == 0L) return(invisible(NULL))
  # enum gpv_status, include/gpvecchia.h:33-44
  msg <- c("1" = "no usable HIP device (the library has no CPU fallback)", "2" = "bad argument",
           "3" = "covType not 'matern' / 'esqe'", "4" = "Matern smoothness outside (0, 60]", "5" = "m + 1 > 192",
           "6" = "HIP runtime error", "7" = "call order", "8" = "neighbour index outside [0, Nlocs]")[as.character(status)]
  stop(sprintf("%s failed with status %d (%s)%s", what, status, msg,
               if (status == 6L) paste0(": ", gpv_last_error()) else ""), call. = FALSE)
}
