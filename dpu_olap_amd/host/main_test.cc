#include <arrow/compute/api.h>

#include "testing/minitest.h"

int main(int argc, char** argv) {
  if (!arrow::compute::Initialize().ok()) return 2;
  return minitest::run_all(argc, argv);
}
